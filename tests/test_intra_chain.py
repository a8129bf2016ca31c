"""The 8x8 CUs of a block as a chain the device runs (include/x265amd.h: x265amd_intra_nxn_job.chain, x265amd_intra_nxn_list), through the
C-ABI: three CUs side by side, each as an NxN evaluation (role 1, decides) and a 2Nx2N evaluation (role 2), queued in advance as two launches on two streams.

The expected results come from the SAME CUs taken one by one through the entry points that have their own parity tests -- x265amd_intra_nxn for either evaluation
(tests/test_intra_nxn.py, tests/test_intra_unit.py: against the oracle), x265amd_intra_cu_bits for both CUs' bits (tests/test_intra_cu_bits.py: against the reference's
checkIntra) -- with the decision on the host: Search::checkIntra's cost (rdcost.h:89-123), checkBestMode's comparison in the order 2Nx2N, NxN (analysis.cpp:3670-3692),
the winner's samples into the picture, its contexts / fraction / directions to the next CU.  What the chain adds is exactly that hand-over, so this is what is compared:
per CU the partitioning, directions, coded block flags, bits, cost, fraction, contexts, levels; at the end the picture."""
import ctypes as C

import numpy as np
import pytest

import hevc_testlib as T
from test_intra_cu_bits import JOB_DT as BITS_JOB_DT, OUT_DT as BITS_OUT_DT

PW = 64                 # the luma picture: 64 x 64, the CUs at y = 16, x = 16, 24, 32
NCU = 3


def chroma_stored(luma_dir, idx):
    lst = [0, 26, 10, 1, 36]
    for i in range(4):
        if luma_dir == lst[i]:
            lst[i] = 34
            break
    return lst[idx]


_STREAMS = []


def _streams():
    import torch
    if not _STREAMS:
        _STREAMS.extend([torch.cuda.Stream(), torch.cuda.Stream()])
    return _STREAMS


def test_chain_records_layout():
    assert T.INTRA_NXN_JOB_DT.fields["chain"][1] == 944 and T.INTRA_NXN_JOB_DT.fields["chain_token"][1] == 1016 and T.INTRA_NXN_JOB_DT.fields["chain_role"][1] == 1024
    assert T.INTRA_CHAIN_DT.fields["ctx"][1] == 16 and T.INTRA_CHAIN_DT.fields["mode"][1] == 176


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [8, 10])
def test_hip_intra_chain_of_three_cus(depth):
    import torch
    H, O = T.load_hip(depth), T.load_oracle(depth)
    lib = H.lib
    en_bits, lps_next = T.entropy_bit_tables()
    lib.x265amd_intra_nxn_list.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    dt = np.uint8 if depth == 8 else np.uint16
    isz = np.dtype(dt).itemsize
    pmax = (1 << depth) - 1
    PEER_BYTES = T.INTRA_NXN_OUT_DT.itemsize + 8 + 160 + 16
    # two streams for the two roles, made once per process: the two launches must run side by side (streams made later may share a hardware queue with these and would take turns --
    # the waits inside the commands are bounded, so that would fail the test with status 2 after two seconds, not hang it)
    s1, s2 = _streams()
    for it in range(6):
        rng = np.random.default_rng(9100 + 10 * depth + it)
        luma = np.clip(np.kron(rng.integers(0, pmax + 1, (8, 8)), np.ones((8, 8), np.int64)) + rng.integers(-9, 10, (PW, PW)) * (1 << (depth - 8)), 0, pmax).astype(dt)
        src = np.clip(luma.astype(np.int64) + rng.integers(-14, 15, (PW, PW)) * (1 << (depth - 8)), 0, pmax).astype(dt)
        chroma = [np.clip(np.kron(rng.integers(0, pmax + 1, (4, 4)), np.ones((8, 8), np.int64)) + rng.integers(-6, 7, (32, 32)), 0, pmax).astype(dt) for _ in range(2)]
        csrc = [np.clip(c.astype(np.int64) + rng.integers(-10, 11, (32, 32)) * (1 << (depth - 8)), 0, pmax).astype(dt) for c in chroma]
        ctx0 = np.zeros(160, np.uint8)
        ctx0[:T.CTX_COUNT] = T.entropy_reset(O, 2, int(rng.integers(22, 38)))
        k = rng.integers(0, T.CTX_COUNT, 30); ctx0[k] = rng.integers(0, 124, 30).astype(np.uint8)
        qp = int(rng.integers(22, 38)) + 6 * (depth - 8)
        prm = dict(strong=1, qp=qp, qpc=max(qp - 1, 6 * (depth - 8)), signhide=1, max_cand=7, frac0=int(rng.integers(0, 1 << 20)))
        prm["lambda"] = int(rng.integers(300, 40000)); prm["lambda2"] = int(rng.integers(20000, 4000000)); prm["psy_scale"] = int(prm["lambda"] * (it % 3) * 128)
        left0 = [int(rng.integers(0, 35)), int(rng.integers(0, 35))]
        above = [[int(rng.integers(0, 35)), int(rng.integers(0, 35))] for _ in range(NCU)]

        def planes():
            return (torch.from_numpy(luma.view(np.uint8).reshape(-1).copy()).cuda(), [torch.from_numpy(c.view(np.uint8).reshape(-1).copy()).cuda() for c in chroma])
        d_src = torch.from_numpy(src.view(np.uint8).reshape(-1).copy()).cuda()
        d_csrc = [torch.from_numpy(c.view(np.uint8).reshape(-1).copy()).cuda() for c in csrc]
        scratch = [dict(cand=torch.zeros(16 * 2048 * isz, dtype=torch.uint8, device="cuda"), cd=torch.zeros(2 * 16 * 1024 * 2, dtype=torch.uint8, device="cuda"),
                        pred=torch.zeros(64 * 64 * isz, dtype=torch.uint8, device="cuda"), layer=torch.zeros(64 * 64 * isz, dtype=torch.uint8, device="cuda"),
                        rect=torch.zeros(64 * 64 * isz, dtype=torch.uint8, device="cuda"), crec=torch.zeros(2 * 32 * 32 * isz, dtype=torch.uint8, device="cuda")) for _ in range(2)]

        def job_of(i, nxn, d_luma, d_ch, ctx, frac, left, abv):
            """the record of CU i (x = 16 + 8 i) coded NxN (role 1's form) or 2Nx2N (role 2's form) as a single, unchained command"""
            s = scratch[0 if nxn else 1]
            x0, y0 = 16 + 8 * i, 16
            j = np.zeros(1, T.INTRA_NXN_JOB_DT)
            units = 4 if nxn else 1
            n, lg = (4, 2) if nxn else (8, 3)
            for u in range(units):
                ux, uy = x0 + 4 * (u & 1), y0 + 4 * (u >> 1)
                j[0]["tmpl"][u]["tu"] = (d_src.data_ptr() + (uy * PW + ux) * isz, s["cand"].data_ptr() + 1024 * isz, s["cd"].data_ptr(), s["cd"].data_ptr() + 16 * 1024 * 2, s["cand"].data_ptr(),
                                         PW, n, n, n, lg, 0, 1, 0, 2, prm["qp"], prm["signhide"], 0)
                j[0]["tmpl"][u]["nb"] = d_luma.data_ptr() + (uy * PW + ux) * isz
                j[0]["tmpl"][u]["avail"] = [0b11111, 0b11110, 0b11110, 0b01110][u] if nxn else 0b111111100       # the 8x8 unit: no below-left (the row below is not coded)
                j[0]["tmpl"][u]["nb_stride"], j[0]["tmpl"][u]["strong"] = PW, prm["strong"]
                off = ((4 * (u >> 1)) * 64 + 4 * (u & 1)) * isz
                j[0]["pred_dst"][u], j[0]["layer_dst"][u], j[0]["recon_dst"][u] = s["pred"].data_ptr() + off, s["layer"].data_ptr() + off, s["rect"].data_ptr() + off
            for pl in range(2):
                j[0]["ctmpl"][pl]["tu"] = (d_csrc[pl].data_ptr() + ((y0 // 2) * 32 + x0 // 2) * isz, 0, s["cd"].data_ptr(), s["cd"].data_ptr() + 16 * 1024 * 2, s["cand"].data_ptr(), 32, 4, 4, 4, 2, 1 + pl, 1, 0, 2,
                                           prm["qpc"], prm["signhide"], 0)
                j[0]["ctmpl"][pl]["nb"] = d_ch[pl].data_ptr() + ((y0 // 2) * 32 + x0 // 2) * isz
                j[0]["ctmpl"][pl]["avail"], j[0]["ctmpl"][pl]["nb_stride"], j[0]["ctmpl"][pl]["strong"] = 0b11110, 32, prm["strong"]
                j[0]["crecon_dst"][pl] = s["crec"].data_ptr() + pl * 32 * 32 * isz
            for f in ("lambda", "lambda2", "psy_scale", "max_cand"):
                j[0][f] = prm[f]
            sf = frac & 32767
            j[0]["scan_frac"] = sf
            j[0]["frac_start"] = [sf + en_bits[int(ctx[8]) ^ (0 if nxn else 1)], sf, sf, sf]         # the partition-size bin in front of the first unit (I slice)
            j[0]["left_mode"], j[0]["above_mode"] = left, abv
            j[0]["slot_pixels"], j[0]["slot_coeffs"], j[0]["ctx"], j[0]["do_chroma"] = 2048, 1024, ctx, 1
            j[0]["reserved"] = (units, lg)
            j[0]["pad"] = (0 if nxn else 1, 0, 0, 0)            # the 2Nx2N evaluation leaves the picture alone
            return j

        def run_single(j):
            d_job = torch.from_numpy(j.view(np.uint8).copy()).cuda()
            d_out = torch.zeros(T.INTRA_NXN_OUT_DT.itemsize, dtype=torch.uint8, device="cuda")
            assert lib.x265amd_intra_nxn(None, C.c_void_p(d_job.data_ptr()), C.c_void_p(d_out.data_ptr())) == 0, lib.x265amd_last_error()
            torch.cuda.synchronize()
            return d_out.cpu().numpy().view(T.INTRA_NXN_OUT_DT)[0].copy()

        def cu_bits(nxn, ctx, frac, out, preds):
            """x265amd_intra_cu_bits on the decided CU -> (frac, mv_frac, ctx)"""
            lev = np.zeros(96, np.int16)
            lev[:64] = out["levels"].reshape(-1)
            lev[64:80], lev[80:96] = out["clevels"][0], out["clevels"][1]
            d_lev = torch.from_numpy(lev).cuda()
            b = np.zeros(1, BITS_JOB_DT)
            b[0]["ctx"], b[0]["frac_bits"] = ctx, frac
            b[0]["log2_cu"], b[0]["nxn"], b[0]["code_part_size"], b[0]["sign_hide"] = 3, int(nxn), 1, prm["signhide"]
            b[0]["chroma_dir"] = chroma_stored(int(out["mode"][0]), int(out["chroma_best"]))
            b[0]["cbf_u"], b[0]["cbf_v"] = int(out["cres"][0]["num_sig"] != 0), int(out["cres"][1]["num_sig"] != 0)
            for u in range(4 if nxn else 1):
                b[0]["luma_dir"][u] = int(out["mode"][u]); b[0]["cbf_y"][u] = int(out["res"][u]["num_sig"] != 0); b[0]["preds"][u] = preds[u]
                b[0]["lev_y"][u] = d_lev.data_ptr() + 2 * 16 * u
            b[0]["lev_u"], b[0]["lev_v"] = d_lev.data_ptr() + 2 * 64, d_lev.data_ptr() + 2 * 80
            d_b = torch.from_numpy(b.view(np.uint8).copy()).cuda()
            d_o = torch.zeros(BITS_OUT_DT.itemsize, dtype=torch.uint8, device="cuda")
            assert lib.x265amd_intra_cu_bits(None, C.c_void_p(d_b.data_ptr()), 1, C.c_void_p(d_o.data_ptr())) == 0, lib.x265amd_last_error()
            torch.cuda.synchronize()
            o = d_o.cpu().numpy().view(BITS_OUT_DT)[0]
            return int(o["frac_bits"]), int(o["mv_frac"]), o["ctx"].copy()

        def cost(dist, bits, energy):
            return dist + ((prm["psy_scale"] * energy) >> 24) + ((bits * prm["lambda2"]) >> 8) if prm["psy_scale"] else dist + ((bits * prm["lambda2"] + 128) >> 8)

        # ---- one by one, the decision on the host ----
        d_luma, d_ch = planes()
        ctx, frac = ctx0.copy(), prm["frac0"]
        left = list(left0)
        want = []
        for i in range(NCU):
            luma_before = d_luma.clone()
            oN = run_single(job_of(i, True, d_luma, d_ch, ctx, frac, left, above[i]))
            recN = d_luma.cpu().numpy().view(dt).reshape(PW, PW)[16:24, 16 + 8 * i:24 + 8 * i].copy()
            crecN = scratch[0]["crec"].cpu().numpy().view(dt).reshape(2, 32, 32)[:, :4, :4].copy()
            d_luma.copy_(luma_before)                               # the chroma blocks in the picture are outside every later CU's neighbourhood: left as they are
            o2 = run_single(job_of(i, False, d_luma, d_ch, ctx, frac, left, above[i]))
            rec2 = scratch[1]["rect"].cpu().numpy().view(dt).reshape(64, 64)[:8, :8].copy()
            crec2 = scratch[1]["crec"].cpu().numpy().view(dt).reshape(2, 32, 32)[:, :4, :4].copy()
            predsN = []
            for u in range(4):
                lm = int(oN["mode"][u - 1]) if u & 1 else left[u >> 1]
                am = int(oN["mode"][u - 2]) if u & 2 else above[i][u & 1]
                predsN.append(T.luma_mpm(lm, am))
            fN, mvN, cN = cu_bits(True, ctx, frac, oN, predsN)
            f2, mv2, c2 = cu_bits(False, ctx, frac, o2, [predsN[0]] * 4)
            distN = sum(int(oN["res"][u]["nz_dist"]) for u in range(4)) + int(oN["cres"][0]["nz_dist"]) + int(oN["cres"][1]["nz_dist"])
            dist2 = int(o2["res"][0]["nz_dist"]) + int(o2["cres"][0]["nz_dist"]) + int(o2["cres"][1]["nz_dist"])
            eN = int(oN["psy_energy"]) if prm["psy_scale"] else 0
            e2 = int(o2["res"][0]["nz_energy"]) if prm["psy_scale"] else 0
            costN, cost2 = cost(distN, fN >> 15, eN), cost(dist2, f2 >> 15, e2)
            nxn = costN < cost2
            o, f, mv, c, rec, crec = (oN, fN, mvN, cN, recN, crecN) if nxn else (o2, f2, mv2, c2, rec2, crec2)
            dirs = [int(o["mode"][u if nxn else 0]) for u in range(4)]
            want.append(dict(part=3 if nxn else 0, dirs=dirs, chroma=chroma_stored(int(o["mode"][0]), int(o["chroma_best"])),
                             cbf_y=[int(o["res"][u if nxn else 0]["num_sig"] != 0) for u in range(4)], cbf_u=int(o["cres"][0]["num_sig"] != 0), cbf_v=int(o["cres"][1]["num_sig"] != 0),
                             rd=min(costN, cost2) if nxn else cost2, other=cost2 if nxn else costN, bits=f >> 15, mv=mv >> 15, frac=f, ctx=c,
                             levels=np.concatenate([o["levels"].reshape(-1), o["clevels"][0], o["clevels"][1]])))
            pic = d_luma.cpu().numpy().view(dt).reshape(PW, PW).copy()
            pic[16:24, 16 + 8 * i:24 + 8 * i] = rec
            d_luma.copy_(torch.from_numpy(pic.view(np.uint8).reshape(-1)).cuda())
            for pl in range(2):
                cp = d_ch[pl].cpu().numpy().view(dt).reshape(32, 32).copy()
                cp[8:12, 8 + 4 * i:12 + 4 * i] = crec[pl]
                d_ch[pl].copy_(torch.from_numpy(cp.view(np.uint8).reshape(-1)).cuda())
            ctx = np.zeros(160, np.uint8); ctx[:] = c
            frac = f
            left = [dirs[1], dirs[3]]
        want_luma = d_luma.cpu().numpy().view(dt).reshape(PW, PW).copy()
        want_ch = [d.cpu().numpy().view(dt).reshape(32, 32).copy() for d in d_ch]

        # ---- the same three CUs as a chain: two launches on two streams, nothing in between ----
        d_luma, d_ch = planes()
        d_chain = torch.zeros(T.INTRA_CHAIN_DT.itemsize, dtype=torch.uint8, device="cuda")
        d_peer = torch.zeros(NCU * PEER_BYTES, dtype=torch.uint8, device="cuda")
        d_res = torch.zeros(NCU * T.INTRA_CU8_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda")
        d_win = torch.zeros(64 * 64 * isz + 2 * 32 * 32 * isz, dtype=torch.uint8, device="cuda")
        jobs = np.zeros(2 * NCU, T.INTRA_NXN_JOB_DT)
        token0 = 1000 + 10 * it
        d_chain.view(torch.int64)[0] = token0                  # the chain's count before the first CU
        for i in range(NCU):
            for role in (1, 2):
                j = job_of(i, role == 1, d_luma, d_ch, ctx0, prm["frac0"], list(left0), above[i])
                j[0]["chain"], j[0]["peer"], j[0]["cu_out"] = d_chain.data_ptr(), d_peer.data_ptr() + i * PEER_BYTES, d_res.data_ptr() + i * T.INTRA_CU8_RESULT_DT.itemsize
                j[0]["chain_token"], j[0]["chain_role"], j[0]["chain_first"], j[0]["chain_index"] = token0 + i, role, int(i == 0), i
                j[0]["mode_src"] = [((i - 1) << 2) | 1, ((i - 1) << 2) | 3, 0xFF, 0xFF] if i else [0xFF] * 4
                if role == 1:
                    j[0]["peer_recon"] = [scratch[1]["rect"].data_ptr(), scratch[1]["crec"].data_ptr(), scratch[1]["crec"].data_ptr() + 32 * 32 * isz]
                    j[0]["win_dst"] = [d_win.data_ptr() + 8 * i * isz, d_win.data_ptr() + (64 * 64 + 4 * i) * isz, d_win.data_ptr() + (64 * 64 + 32 * 32 + 4 * i) * isz]
                jobs[2 * i + role - 1] = j[0]
        d_jobs = torch.from_numpy(jobs.view(np.uint8).copy()).cuda()
        d_out = torch.zeros(T.INTRA_NXN_OUT_DT.itemsize, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        stride = 2 * T.INTRA_NXN_JOB_DT.itemsize
        assert lib.x265amd_intra_nxn_list(C.c_void_p(s2.cuda_stream), C.c_void_p(d_jobs.data_ptr() + T.INTRA_NXN_JOB_DT.itemsize), NCU, stride, C.c_void_p(d_out.data_ptr())) == 0
        assert lib.x265amd_intra_nxn_list(C.c_void_p(s1.cuda_stream), C.c_void_p(d_jobs.data_ptr()), NCU, stride, C.c_void_p(d_out.data_ptr())) == 0
        torch.cuda.synchronize()
        res = d_res.cpu().numpy().view(T.INTRA_CU8_RESULT_DT)
        for i in range(NCU):
            r, w = res[i], want[i]
            tag = "depth %d case %d CU %d" % (depth, it, i)
            assert int(r["status"]) == 1, tag
            got = dict(part=int(r["part_size"]), dirs=[int(v) for v in r["luma_dir"]], chroma=int(r["chroma_dir"]), cbf_y=[int(v) for v in r["cbf_y"]], cbf_u=int(r["cbf_u"]), cbf_v=int(r["cbf_v"]),
                       rd=int(r["rd_cost"]), other=int(r["other_cost"]), bits=int(r["total_bits"]), mv=int(r["mv_bits"]), frac=int(r["frac_bits"]))
            for f in got:
                assert got[f] == w[f], "%s: %s %s, one by one %s" % (tag, f, got[f], w[f])
            assert np.array_equal(r["ctx"][:T.CTX_COUNT], w["ctx"][:T.CTX_COUNT]), tag + ": contexts"
            assert np.array_equal(r["levels"], w["levels"]), tag + ": levels"
        assert np.array_equal(d_luma.cpu().numpy().view(dt).reshape(PW, PW), want_luma), "depth %d case %d: the picture" % (depth, it)
        for pl in range(2):
            assert np.array_equal(d_ch[pl].cpu().numpy().view(dt).reshape(32, 32), want_ch[pl]), "depth %d case %d: chroma plane %d" % (depth, it, pl)
        win = d_win.cpu().numpy().view(dt)
        assert np.array_equal(win[:64 * 64].reshape(64, 64)[:8, :8 * NCU], want_luma[16:24, 16:16 + 8 * NCU]), "the winners' samples beside the picture"
