"""x265amd_intra_nxn with ONE unit of 8x8, 16x16 or 32x32 (include/x265amd.h: num_units 1, unit_log2 3..5): a CU coded 2Nx2N decided on the device with its chroma decision
(blocks of half the size), as the I pictures' CUs use it (Search::estIntraPredQT + estIntraPredChromaQT, search.cpp:1509-1696, :1754-1889) -- and with pick_sa8d the intra try
of an inter picture's CU (checkIntraInInter + encodeIntraInInter, search.cpp:1291-1507): the mode with the least SA8D cost in the order DC, planar, angular instead of a
candidate list.  no_picture leaves the reconstructed planes alone.  The expected result is assembled from the oracle's pieces as in test_intra_nxn.py."""
import ctypes as C

import numpy as np
import pytest

import hevc_testlib as T
from test_intra_nxn import rd_cost

_ptr = lambda a: a.ctypes.data_as(C.c_void_p)
PW, CU0 = 128, 32          # luma plane 128 x 128 with the CU at (32, 32); chroma planes 64 x 64 with the block at (16, 16)


def expected(O, luma, fenc, chroma, cfenc, ctx, prm, en_bits, lps_next):
    log2, N = prm["log2"], 1 << prm["log2"]
    adi = int(ctx[13])
    rbits = ((prm["scan_frac"] + en_bits[adi ^ 0]) >> 15) + 5
    mpm_base = (prm["scan_frac"] + en_bits[adi ^ 1]) >> 15
    preds = T.luma_mpm(prm["left_mode"][0], prm["above_mode"][0])
    ones = np.ones(4 * (N // 4) + 1, np.uint8)            # one flag per four neighbouring samples: below-left, left, above-left, above, above-right -- all there
    case = dict(plane=np.ascontiguousarray(luma).ravel(), stride=PW, off=CU0 * PW + CU0, log2=log2, flags=ones, strong=prm["strong"], fenc=np.ascontiguousarray(fenc), ttype=0,
                slice=prm["slice"], qp=prm["qp"], signhide=prm["signhide"], rdoq=0, tudepth=0, psyrdoq=0, ctx=ctx[:T.CTX_COUNT])
    sa8d = T.intra_run_host(O, [case])[0][2]
    if prm["pick"]:
        def cost_of(m):
            b = rbits
            if m in preds:
                b = mpm_base + (1 if m == preds[0] else 2)
            return int(sa8d[m]) + ((b * prm["lambda"] + 128) >> 8)
        bm = 1
        for m in [0] + list(range(2, 35)):
            if cost_of(m) < cost_of(bm):
                bm = m
        modes = [bm]
    else:
        modes = T.intra_pu_candidates(sa8d, preds, rbits, mpm_base, prm["lambda"], prm["max_cand"])
    per = T.intra_tu_run_host(O, [dict(case, mode=m) for m in modes])
    best, bi = None, -1
    for i, (m, (st, pred, recon, coeff, resi)) in enumerate(zip(modes, per)):
        frac = prm["frac_start"]
        pidx = preds.index(m) if m in preds else -1
        frac += en_bits[adi ^ (1 if pidx != -1 else 0)]
        frac += (1 + (pidx != 0) if pidx != -1 else 5) << 15
        frac += en_bits[int(ctx[29]) ^ (1 if st[0] else 0)]                 # C_QT_CBF + 1: transform depth 0
        if st[0]:
            cc = dict(ctx=ctx[:T.CTX_COUNT], log2=log2, ttype=0, intra=1, dir=m, signhide=prm["signhide"])
            frac += T.coeff_bits_run(O, [cc], [(st[0], coeff)])[0][0]
        cost = rd_cost(st[3], frac >> 15, st[4] if prm["psy_scale"] else 0, prm["lambda2"], prm["psy_scale"])
        if best is None or cost < best:
            best, bi = cost, i
    st, pred, recon, coeff, resi = per[bi]
    win = modes[bi]
    lst = [0, 26, 10, 1, 36]
    for i in range(4):
        if win == lst[i]:
            lst[i] = 34
            break
    cn, clog2 = N // 2, log2 - 1
    cbest, ck, cres, clev, crec, last = None, -1, None, None, None, None
    for k, listed in enumerate(lst):
        mode = win if listed == 36 else listed
        cw = np.array(ctx[:T.CTX_COUNT], np.uint8)
        frac = prm["scan_frac"] + en_bits[int(cw[14]) ^ (0 if listed == 36 else 1)]
        cw[14] = T.cabac_next_state(int(cw[14]), 0 if listed == 36 else 1, lps_next)
        if listed != 36:
            frac += 2 << 15
        cper = []
        for pl in range(2):
            ccase = dict(plane=np.ascontiguousarray(chroma[pl]).ravel(), stride=PW // 2, off=(CU0 // 2) * (PW // 2) + CU0 // 2, log2=clog2, flags=np.ones(4 * (cn // 4) + 1, np.uint8), strong=prm["strong"],
                         fenc=np.ascontiguousarray(cfenc[pl]), ttype=1 + pl, mode=mode, slice=prm["slice"], qp=prm["qpc"], signhide=prm["signhide"], rdoq=0, tudepth=0, psyrdoq=0,
                         ctx=ctx[:T.CTX_COUNT])
            cper.append(T.intra_tu_run_host(O, [ccase])[0])
        for pl in range(2):
            v = 1 if cper[pl][0][0] else 0
            frac += en_bits[int(cw[30]) ^ v]
            cw[30] = T.cabac_next_state(int(cw[30]), v, lps_next)
        dist = energy = 0
        for pl in range(2):
            cst, cpred, crecon, ccoeff, cresi = cper[pl]
            if cst[0]:
                cc = dict(ctx=cw.copy(), log2=clog2, ttype=1 + pl, intra=1, dir=mode, signhide=prm["signhide"])
                b, newctx = T.coeff_bits_run(O, [cc], [(cst[0], ccoeff)])[0]
                frac += b
                cw = np.array(newctx, np.uint8)
            dist += cst[3]; energy += cst[4]
        cost = rd_cost(dist, frac >> 15, energy if prm["psy_scale"] else 0, prm["lambda2"], prm["psy_scale"])
        if cbest is None or cost < cbest:
            cbest, ck, cres, clev, crec = cost, k, [cper[0][0], cper[1][0]], [cper[0][3].copy(), cper[1][3].copy()], [cper[0][2], cper[1][2]]
        last = [cper[0][2], cper[1][2]]
    return dict(mode=win, res=st, levels=coeff.copy(), recon=recon, pred=pred, sa8d=int(sa8d[win]), chroma_best=ck, cres=cres, clevels=clev, crec=crec, clast=last)


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [8, 10])
def test_hip_intra_single_unit(depth):
    import torch
    H, O = T.load_hip(depth), T.load_oracle(depth)
    en_bits, lps_next = T.entropy_bit_tables()
    rng = np.random.default_rng(5150 + depth)
    dt = np.uint8 if depth == 8 else np.uint16
    pmax = (1 << depth) - 1
    isz = np.dtype(dt).itemsize
    for it in range(18):
        log2 = 3 + it % 3
        N, cn = 1 << log2, (1 << log2) // 2
        pick, nopic = (it // 3) & 1, (it // 6) & 1
        luma = np.clip(np.kron(rng.integers(0, pmax + 1, (8, 8)), np.ones((16, 16), np.int64)) + rng.integers(-9, 10, (PW, PW)) * (1 << (depth - 8)), 0, pmax).astype(dt)
        fenc = np.clip(luma[CU0:CU0 + N, CU0:CU0 + N].astype(np.int64) + rng.integers(-14, 15, (N, N)) * (1 << (depth - 8)), 0, pmax).astype(dt)
        chroma = [np.clip(np.kron(rng.integers(0, pmax + 1, (4, 4)), np.ones((16, 16), np.int64)) + rng.integers(-6, 7, (PW // 2, PW // 2)), 0, pmax).astype(dt) for _ in range(2)]
        cfenc = [np.clip(c[CU0 // 2:CU0 // 2 + cn, CU0 // 2:CU0 // 2 + cn].astype(np.int64) + rng.integers(-10, 11, (cn, cn)) * (1 << (depth - 8)), 0, pmax).astype(dt) for c in chroma]
        slice_type = int(rng.integers(0, 3))
        ctx160 = np.zeros(160, np.uint8)
        ctx160[:T.CTX_COUNT] = T.entropy_reset(O, slice_type, int(rng.integers(20, 40)))
        if it & 1:
            k = rng.integers(0, T.CTX_COUNT, 30); ctx160[k] = rng.integers(0, 124, 30).astype(np.uint8)
        qp = int(rng.integers(18, 40)) + 6 * (depth - 8)
        prm = dict(log2=log2, pick=pick, strong=int(rng.integers(0, 2)), slice=slice_type, qp=qp, qpc=max(qp - int(rng.integers(0, 4)), 6 * (depth - 8)), signhide=int(rng.integers(0, 2)),
                   scan_frac=int(rng.integers(0, 32768)), frac_start=int(rng.integers(0, 200000)), left_mode=[int(rng.integers(0, 35)), 1], above_mode=[int(rng.integers(0, 35)), 1],
                   max_cand=1 if pick else int(rng.integers(3, 8)))
        prm["lambda"] = int(rng.integers(300, 40000)); prm["lambda2"] = int(rng.integers(2000, 4000000)); prm["psy_scale"] = int(prm["lambda"] * rng.integers(0, 3) * 128)
        want = expected(O, luma, fenc, chroma, cfenc, ctx160, prm, en_bits, lps_next)

        d_luma = torch.from_numpy(luma.view(np.uint8).copy()).cuda(); d_fenc = torch.from_numpy(fenc.view(np.uint8).copy()).cuda()
        d_ch = [torch.from_numpy(c.view(np.uint8).copy()).cuda() for c in chroma]; d_cf = [torch.from_numpy(c.view(np.uint8).copy()).cuda() for c in cfenc]
        d_cand = torch.zeros(16 * 2048 * isz, dtype=torch.uint8, device="cuda"); d_cd = torch.zeros(2 * 16 * 1024 * 2, dtype=torch.uint8, device="cuda")
        d_pred = torch.zeros(64 * 64 * isz, dtype=torch.uint8, device="cuda"); d_layer = torch.zeros(64 * 64 * isz, dtype=torch.uint8, device="cuda")
        d_rect = torch.zeros(64 * 64 * isz, dtype=torch.uint8, device="cuda"); d_crec = torch.zeros(2 * 32 * 32 * isz, dtype=torch.uint8, device="cuda")
        d_lev = torch.zeros(1024, dtype=torch.int16, device="cuda"); d_clev = torch.zeros(2 * 256, dtype=torch.int16, device="cuda")
        job = np.zeros(1, T.INTRA_NXN_JOB_DT)
        allav = (1 << (4 * (N // 4) + 1)) - 1
        job[0]["tmpl"][0]["tu"] = (d_fenc.data_ptr(), d_cand.data_ptr() + 1024 * isz, d_cd.data_ptr(), d_cd.data_ptr() + 16 * 1024 * 2, d_cand.data_ptr(), N, N, N, N, log2, 0, 1, 0,
                                   prm["slice"], prm["qp"], prm["signhide"], 0)
        job[0]["tmpl"][0]["nb"] = d_luma.data_ptr() + (CU0 * PW + CU0) * isz
        job[0]["tmpl"][0]["avail"], job[0]["tmpl"][0]["nb_stride"], job[0]["tmpl"][0]["strong"] = allav, PW, prm["strong"]
        job[0]["pred_dst"][0], job[0]["layer_dst"][0], job[0]["recon_dst"][0] = d_pred.data_ptr(), d_layer.data_ptr(), d_rect.data_ptr()
        callav = (1 << (4 * (cn // 4) + 1)) - 1
        for pl in range(2):
            job[0]["ctmpl"][pl]["tu"] = (d_cf[pl].data_ptr(), 0, d_cd.data_ptr(), d_cd.data_ptr() + 16 * 1024 * 2, d_cand.data_ptr(), cn, cn, cn, cn, log2 - 1, 1 + pl, 1, 0, prm["slice"],
                                         prm["qpc"], prm["signhide"], 0)
            job[0]["ctmpl"][pl]["nb"] = d_ch[pl].data_ptr() + ((CU0 // 2) * (PW // 2) + CU0 // 2) * isz
            job[0]["ctmpl"][pl]["avail"], job[0]["ctmpl"][pl]["nb_stride"], job[0]["ctmpl"][pl]["strong"] = callav, PW // 2, prm["strong"]
            job[0]["crecon_dst"][pl] = d_crec.data_ptr() + pl * 32 * 32 * isz
        for f in ("lambda", "lambda2", "psy_scale", "scan_frac", "max_cand"):
            job[0][f] = prm[f]
        job[0]["frac_start"][0] = prm["frac_start"]
        job[0]["left_mode"], job[0]["above_mode"] = prm["left_mode"], prm["above_mode"]
        job[0]["slot_pixels"], job[0]["slot_coeffs"], job[0]["ctx"], job[0]["do_chroma"] = 2048, 1024, ctx160, 1
        job[0]["reserved"] = (1, log2)                      # num_units, unit_log2
        job[0]["pad"] = (nopic, pick, 0, 0)                 # no_picture, pick_sa8d
        if log2 > 3:
            job[0]["levels_dst"], job[0]["clevels_dst"] = d_lev.data_ptr(), d_clev.data_ptr()
        d_job = torch.from_numpy(job.view(np.uint8).copy()).cuda()
        d_out = torch.zeros(T.INTRA_NXN_OUT_DT.itemsize, dtype=torch.uint8, device="cuda")
        assert H.lib.x265amd_intra_nxn(None, C.c_void_p(d_job.data_ptr()), C.c_void_p(d_out.data_ptr())) == 0
        torch.cuda.synchronize()
        out = d_out.cpu().numpy().view(T.INTRA_NXN_OUT_DT)[0]
        tag = (it, log2, pick, nopic)
        assert int(out["mode"][0]) == want["mode"], (tag, int(out["mode"][0]), want["mode"])
        r = out["res"][0]
        assert (int(r["num_sig"]), int(r["zero_dist"]), int(r["zero_energy"]), int(r["nz_dist"]), int(r["nz_energy"])) == tuple(int(v) for v in want["res"]), (tag, "result")
        lev = d_lev.cpu().numpy()[:N * N] if log2 > 3 else out["levels"].reshape(-1)[:N * N]
        assert np.array_equal(lev, want["levels"]), (tag, "levels")
        if pick:
            assert int(out["chroma_reserved"]) == want["sa8d"], (tag, "SA8D of the picked mode")
        got_luma = d_luma.cpu().numpy().view(dt).reshape(PW, PW)
        exp_luma = luma.copy()
        if not nopic:
            exp_luma[CU0:CU0 + N, CU0:CU0 + N] = want["recon"]
        assert np.array_equal(got_luma, exp_luma), (tag, "the reconstructed plane")
        assert np.array_equal(d_layer.cpu().numpy().view(dt).reshape(64, 64)[:N, :N], want["recon"]), (tag, "layer tile")
        assert np.array_equal(d_rect.cpu().numpy().view(dt).reshape(64, 64)[:N, :N], want["recon"]), (tag, "reconstruction tile")
        assert np.array_equal(d_pred.cpu().numpy().view(dt).reshape(64, 64)[:N, :N], want["pred"]), (tag, "prediction tile")
        assert int(out["chroma_best"]) == want["chroma_best"], (tag, "chroma mode", int(out["chroma_best"]), want["chroma_best"])
        clev = d_clev.cpu().numpy() if cn > 4 else out["clevels"].reshape(-1)
        for pl in range(2):
            r = out["cres"][pl]
            assert (int(r["num_sig"]), int(r["zero_dist"]), int(r["zero_energy"]), int(r["nz_dist"]), int(r["nz_energy"])) == tuple(int(v) for v in want["cres"][pl]), (tag, "chroma result", pl)
            assert np.array_equal(clev[pl * cn * cn:(pl + 1) * cn * cn], want["clevels"][pl]), (tag, "chroma levels", pl)
            assert np.array_equal(d_crec.cpu().numpy().view(dt).reshape(2, 32, 32)[pl, :cn, :cn], want["crec"][pl]), (tag, "chroma reconstruction tile", pl)
            exp_c = chroma[pl].copy()
            if not nopic:
                exp_c[CU0 // 2:CU0 // 2 + cn, CU0 // 2:CU0 // 2 + cn] = want["clast"][pl]
            assert np.array_equal(d_ch[pl].cpu().numpy().view(dt).reshape(PW // 2, PW // 2), exp_c), (tag, "the chroma plane", pl)
