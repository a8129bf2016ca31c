"""x265amd_intra_nxn (include/x265amd.h): an 8x8 CU coded NxN as ONE launch with the decisions made on the device.  The expected result is assembled from the oracle's
pieces in the reference's order (Search::estIntraPredQT, search.cpp:1509-1696; codeIntraLumaQT :357-400; estIntraPredChromaQT :1754-1889): per 4x4 unit the most probable
modes, the oracle's 35-mode scan on the current reconstruction, the candidate list, the oracle's intra TU step per candidate, the bits (prev_intra_luma_pred_flag, mode index, coded
block flag, the oracle's bits-only coefficient coding) and the cost; the winner's reconstruction becomes the next unit's neighbourhood.  Then the CU's luma measurements and the
chroma decision the same way."""
import ctypes as C

import numpy as np
import pytest

import hevc_testlib as T

_ptr = lambda a: a.ctypes.data_as(C.c_void_p)


def rd_cost(dist, bits, energy, lambda2, psy_scale):
    return dist + ((psy_scale * energy) >> 24) + ((bits * lambda2) >> 8) if psy_scale else dist + ((bits * lambda2 + 128) >> 8)


def expected(O, depth, luma, fenc, chroma, cfenc, ctx, prm, en_bits, lps_next):
    """luma: 2-D reconstruction plane with the CU at (16, 16); chroma: two planes with the CU's 4x4 block at (8, 8).  Returns what x265amd_intra_nxn must deliver."""
    luma = luma.copy()
    stride = luma.shape[1]
    win, res, lev = [], [], []
    flags_of = [[1, 1, 1, 1, 1], [0, 1, 1, 1, 1], [0, 1, 1, 1, 1], [0, 1, 1, 1, 0]]          # below-left, left, above-left, above, above-right per unit
    adi = int(ctx[13])
    rbits = ((prm["scan_frac"] + en_bits[adi ^ 0]) >> 15) + 5
    mpm_base = (prm["scan_frac"] + en_bits[adi ^ 1]) >> 15
    preds_all = []
    for k in range(4):
        y0, x0 = 16 + 4 * (k >> 1), 16 + 4 * (k & 1)
        left = win[k - 1] if k & 1 else prm["left_mode"][k >> 1]
        above = win[k - 2] if k & 2 else prm["above_mode"][k & 1]
        preds = T.luma_mpm(left, above)
        preds_all.append(preds)
        case = dict(plane=np.ascontiguousarray(luma).ravel(), stride=stride, off=y0 * stride + x0, log2=2, flags=np.array(flags_of[k], np.uint8), strong=prm["strong"],
                    fenc=np.ascontiguousarray(fenc[4 * (k >> 1):4 * (k >> 1) + 4, 4 * (k & 1):4 * (k & 1) + 4]), ttype=0, slice=prm["slice"], qp=prm["qp"], signhide=prm["signhide"],
                    rdoq=prm.get("rdoq", 0), tudepth=1, psyrdoq=prm.get("psyrdoq", 0), ctx=ctx[:T.CTX_COUNT])
        sa8d = T.intra_run_host(O, [case])[0][2]
        modes = T.intra_pu_candidates(sa8d, preds, rbits, mpm_base, prm["lambda"], prm["max_cand"])
        per = T.intra_tu_run_host(O, [dict(case, mode=m) for m in modes])
        best, bi = None, -1
        for i, (m, (st, pred, recon, coeff, resi)) in enumerate(zip(modes, per)):
            frac = prm["frac_start"][k]
            pidx = preds.index(m) if m in preds else -1
            frac += en_bits[adi ^ (1 if pidx != -1 else 0)]
            frac += (1 + (pidx != 0) if pidx != -1 else 5) << 15
            frac += en_bits[int(ctx[28]) ^ (1 if st[0] else 0)]
            if st[0]:
                cc = dict(ctx=ctx[:T.CTX_COUNT], log2=2, ttype=0, intra=1, dir=m, signhide=prm["signhide"])
                frac += T.coeff_bits_run(O, [cc], [(st[0], coeff)])[0][0]
            cost = rd_cost(st[3], frac >> 15, st[4] if prm["psy_scale"] else 0, prm["lambda2"], prm["psy_scale"])
            if best is None or cost < best:
                best, bi = cost, i
        st, pred, recon, coeff, resi = per[bi]
        win.append(modes[bi]); res.append(st); lev.append(coeff.copy())
        luma[y0:y0 + 4, x0:x0 + 4] = recon
        prm.setdefault("pred_blocks", []).append(pred)
    pred8 = np.zeros((8, 8), luma.dtype)
    for k in range(4):
        pred8[4 * (k >> 1):4 * (k >> 1) + 4, 4 * (k & 1):4 * (k & 1) + 4] = prm["pred_blocks"][k]
    rec8 = np.ascontiguousarray(luma[16:24, 16:24])
    f8 = np.ascontiguousarray(fenc)
    O.lib.orc_psy_cost_pp.restype = C.c_int; O.lib.orc_sse_pp.restype = C.c_uint64
    psy = O.lib.orc_psy_cost_pp(1, _ptr(f8), C.c_int64(8), _ptr(rec8), C.c_int64(8))
    sse = O.lib.orc_sse_pp(1, _ptr(f8), C.c_int64(8), _ptr(pred8), C.c_int64(8))
    # chroma: the five allowed modes
    lst = [0, 26, 10, 1, 36]
    for i in range(4):
        if win[0] == lst[i]:
            lst[i] = 34
            break
    cbest, ck, cres, clev, crec, last = None, -1, None, None, None, None
    for k, listed in enumerate(lst):
        mode = win[0] if listed == 36 else listed
        cw = np.array(ctx[:T.CTX_COUNT], np.uint8)
        frac = prm["scan_frac"] + en_bits[int(cw[14]) ^ (0 if listed == 36 else 1)]
        cw[14] = T.cabac_next_state(int(cw[14]), 0 if listed == 36 else 1, lps_next)
        if listed != 36:
            frac += 2 << 15
        per = []
        for pl in range(2):
            plane = chroma[pl]
            case = dict(plane=np.ascontiguousarray(plane).ravel(), stride=plane.shape[1], off=8 * plane.shape[1] + 8, log2=2, flags=np.array([1, 1, 1, 1, 1], np.uint8), strong=prm["strong"],
                        fenc=np.ascontiguousarray(cfenc[pl]), ttype=1 + pl, mode=mode, slice=prm["slice"], qp=prm["qpc"], signhide=prm["signhide"], rdoq=prm.get("rdoq", 0), tudepth=1,
                        psyrdoq=prm.get("psyrdoq", 0), ctx=ctx[:T.CTX_COUNT])
            per.append(T.intra_tu_run_host(O, [case])[0])
        for pl in range(2):
            v = 1 if per[pl][0][0] else 0
            frac += en_bits[int(cw[30]) ^ v]
            cw[30] = T.cabac_next_state(int(cw[30]), v, lps_next)
        dist = energy = 0
        for pl in range(2):
            st, pred, recon, coeff, resi = per[pl]
            if st[0]:
                cc = dict(ctx=cw.copy(), log2=2, ttype=1 + pl, intra=1, dir=mode, signhide=prm["signhide"])
                b, newctx = T.coeff_bits_run(O, [cc], [(st[0], coeff)])[0]
                frac += b
                cw = np.array(newctx, np.uint8)
            dist += st[3]; energy += st[4]
        cost = rd_cost(dist, frac >> 15, energy if prm["psy_scale"] else 0, prm["lambda2"], prm["psy_scale"])
        if cbest is None or cost < cbest:
            cbest, ck, cres, clev, crec = cost, k, [per[0][0], per[1][0]], [per[0][3].copy(), per[1][3].copy()], [per[0][2], per[1][2]]
        last = [per[0][2], per[1][2]]
    return dict(modes=win, res=res, levels=lev, luma=luma, pred8=pred8, psy=psy, sse=sse, chroma_best=ck, cres=cres, clevels=clev, crec=crec, clast=last)


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [8, 10])
def test_hip_intra_nxn(depth):
    import torch
    H, O = T.load_hip(depth), T.load_oracle(depth)
    en_bits, lps_next = T.entropy_bit_tables()
    rng = np.random.default_rng(4711 + depth)
    dt = np.uint8 if depth == 8 else np.uint16
    pmax = (1 << depth) - 1
    isz = np.dtype(dt).itemsize
    for it in range(24):
        base = rng.integers(0, pmax + 1, (6, 6))
        luma = np.clip(np.kron(base, np.ones((8, 8), np.int64)) + rng.integers(-9, 10, (48, 48)) * (1 << (depth - 8)), 0, pmax).astype(dt)
        fenc = np.clip(luma[16:24, 16:24].astype(np.int64) + rng.integers(-14, 15, (8, 8)) * (1 << (depth - 8)), 0, pmax).astype(dt)
        chroma = [np.clip(np.kron(rng.integers(0, pmax + 1, (3, 3)), np.ones((8, 8), np.int64)) + rng.integers(-6, 7, (24, 24)), 0, pmax).astype(dt) for _ in range(2)]
        cfenc = [np.clip(c[8:12, 8:12].astype(np.int64) + rng.integers(-10, 11, (4, 4)) * (1 << (depth - 8)), 0, pmax).astype(dt) for c in chroma]
        slice_type = int(rng.integers(0, 3))
        ctx160 = np.zeros(160, np.uint8)
        ctx160[:T.CTX_COUNT] = T.entropy_reset(O, slice_type, int(rng.integers(20, 40)))
        if it & 1:
            k = rng.integers(0, T.CTX_COUNT, 30); ctx160[k] = rng.integers(0, 124, 30).astype(np.uint8)
        qp = int(rng.integers(18, 40)) + 6 * (depth - 8)
        prm = dict(strong=int(rng.integers(0, 2)), slice=slice_type, qp=qp, qpc=max(qp - int(rng.integers(0, 4)), 6 * (depth - 8)), signhide=int(rng.integers(0, 2)),
                   scan_frac=int(rng.integers(0, 32768)), frac_start=[int(rng.integers(0, 200000))] + [0, 0, 0], left_mode=[int(rng.integers(0, 35)), int(rng.integers(0, 35))],
                   above_mode=[int(rng.integers(0, 35)), int(rng.integers(0, 35))], max_cand=int(rng.integers(4, 9)))
        prm["lambda"] = int(rng.integers(300, 40000)); prm["lambda2"] = int(rng.integers(2000, 4000000)); prm["psy_scale"] = int(prm["lambda"] * rng.integers(0, 3) * 128)
        for k in range(1, 4):
            prm["frac_start"][k] = prm["scan_frac"]
        # RDOQ (Quant::rdoQuant with the bit estimates of the command's start contexts) in the later cases: levels 1 and 2, with and without psy-rdoq; every third of
        # them in the general form (a wavefront per candidate), the others in the sixteen-lane form
        if it >= 10:
            prm["rdoq"], prm["psyrdoq"] = 1 + (it & 1), int(rng.choice([0, 256, 1024]))
        want = expected(O, depth, luma, fenc, chroma, cfenc, ctx160, dict(prm), en_bits, lps_next)

        d_luma = torch.from_numpy(luma.view(np.uint8).copy()).cuda(); d_fenc = torch.from_numpy(fenc.view(np.uint8).copy()).cuda()
        d_ch = [torch.from_numpy(c.view(np.uint8).copy()).cuda() for c in chroma]; d_cf = [torch.from_numpy(c.view(np.uint8).copy()).cuda() for c in cfenc]
        d_cand = torch.zeros(16 * 2048 * isz, dtype=torch.uint8, device="cuda"); d_cd = torch.zeros(2 * 16 * 1024 * 2, dtype=torch.uint8, device="cuda")
        d_pred = torch.zeros(64 * 64 * isz, dtype=torch.uint8, device="cuda"); d_layer = torch.zeros(64 * 64 * isz, dtype=torch.uint8, device="cuda")
        d_crec = torch.zeros(2 * 32 * 32 * isz, dtype=torch.uint8, device="cuda")
        job = np.zeros(1, T.INTRA_NXN_JOB_DT)
        flags_of = [0b11111, 0b11110, 0b11110, 0b01110]
        for k in range(4):
            y0, x0 = 16 + 4 * (k >> 1), 16 + 4 * (k & 1)
            job[0]["tmpl"][k]["tu"] = (d_fenc.data_ptr() + (4 * (k >> 1) * 8 + 4 * (k & 1)) * isz, d_cand.data_ptr() + 1024 * isz, d_cd.data_ptr(), d_cd.data_ptr() + 16 * 1024 * 2,
                                       d_cand.data_ptr(), 8, 4, 4, 4, 2, 0, 1, 0, prm["slice"], prm["qp"], prm["signhide"], 0)
            job[0]["tmpl"][k]["nb"] = d_luma.data_ptr() + (y0 * 48 + x0) * isz
            job[0]["tmpl"][k]["avail"], job[0]["tmpl"][k]["nb_stride"], job[0]["tmpl"][k]["strong"] = flags_of[k], 48, prm["strong"]
            job[0]["pred_dst"][k] = d_pred.data_ptr() + ((4 * (k >> 1)) * 64 + 4 * (k & 1)) * isz
            job[0]["layer_dst"][k] = d_layer.data_ptr() + ((4 * (k >> 1)) * 64 + 4 * (k & 1)) * isz
        for pl in range(2):
            job[0]["ctmpl"][pl]["tu"] = (d_cf[pl].data_ptr(), 0, d_cd.data_ptr(), d_cd.data_ptr() + 16 * 1024 * 2, d_cand.data_ptr(), 4, 4, 4, 4, 2, 1 + pl, 1, 0, prm["slice"], prm["qpc"],
                                         prm["signhide"], 0)
            job[0]["ctmpl"][pl]["nb"] = d_ch[pl].data_ptr() + (8 * 24 + 8) * isz
            job[0]["ctmpl"][pl]["avail"], job[0]["ctmpl"][pl]["nb_stride"], job[0]["ctmpl"][pl]["strong"] = 0b11111, 24, prm["strong"]
            job[0]["crecon_dst"][pl] = d_crec.data_ptr() + pl * 32 * 32 * isz
        for f in ("lambda", "lambda2", "psy_scale", "scan_frac", "max_cand"):
            job[0][f] = prm[f]
        job[0]["frac_start"], job[0]["left_mode"], job[0]["above_mode"] = prm["frac_start"], prm["left_mode"], prm["above_mode"]
        job[0]["slot_pixels"], job[0]["slot_coeffs"], job[0]["ctx"], job[0]["do_chroma"] = 2048, 1024, ctx160, 1
        if prm.get("rdoq"):
            job[0]["rdoq_level"], job[0]["psy_rdoq_scale"], job[0]["rdoq_tu_depth"], job[0]["rdoq_general"] = prm["rdoq"], prm["psyrdoq"], 1, int(it % 3 == 0)
            for pl, q in enumerate((prm["qp"], prm["qpc"], prm["qpc"])):
                l2, l1 = C.c_int64(0), C.c_int32(0)
                H.lib.x265amd_rdoq_lambda(q, C.byref(l2), C.byref(l1))
                job[0]["rdoq_lambda2"][pl], job[0]["rdoq_lambda"][pl] = l2.value, l1.value
        d_job = torch.from_numpy(job.view(np.uint8).copy()).cuda()
        d_out = torch.zeros(T.INTRA_NXN_OUT_DT.itemsize, dtype=torch.uint8, device="cuda")
        assert H.lib.x265amd_intra_nxn(None, C.c_void_p(d_job.data_ptr()), C.c_void_p(d_out.data_ptr())) == 0
        torch.cuda.synchronize()
        out = d_out.cpu().numpy().view(T.INTRA_NXN_OUT_DT)[0]
        assert [int(m) for m in out["mode"]] == want["modes"], (it, out["mode"], want["modes"])
        for k in range(4):
            r = out["res"][k]
            assert (int(r["num_sig"]), int(r["zero_dist"]), int(r["zero_energy"]), int(r["nz_dist"]), int(r["nz_energy"])) == want["res"][k], (it, k)
            assert np.array_equal(out["levels"][k], want["levels"][k]), (it, k, "levels")
        got_luma = d_luma.cpu().numpy().view(dt).reshape(48, 48)
        assert np.array_equal(got_luma, want["luma"]), (it, "reconstruction in the picture")
        assert np.array_equal(d_layer.cpu().numpy().view(dt).reshape(64, 64)[:8, :8], want["luma"][16:24, 16:24]), (it, "layer tile")
        assert np.array_equal(d_pred.cpu().numpy().view(dt).reshape(64, 64)[:8, :8], want["pred8"]), (it, "prediction tile")
        assert int(out["psy_energy"]) == want["psy"] and int(out["res_energy"]) == want["sse"], (it, "luma measurements")
        assert int(out["chroma_best"]) == want["chroma_best"], (it, "chroma mode", int(out["chroma_best"]), want["chroma_best"])
        for pl in range(2):
            r = out["cres"][pl]
            assert (int(r["num_sig"]), int(r["zero_dist"]), int(r["zero_energy"]), int(r["nz_dist"]), int(r["nz_energy"])) == want["cres"][pl], (it, "chroma result", pl)
            assert np.array_equal(out["clevels"][pl], want["clevels"][pl]), (it, "chroma levels", pl)
            assert np.array_equal(d_crec.cpu().numpy().view(dt).reshape(2, 32, 32)[pl, :4, :4], want["crec"][pl]), (it, "chroma reconstruction tile", pl)
            assert np.array_equal(d_ch[pl].cpu().numpy().view(dt).reshape(24, 24)[8:12, 8:12], want["clast"][pl]), (it, "the picture keeps the last tried mode's chroma", pl)
