"""Lookahead lowres pipeline, first stage: x265amd_lowres_init and x265amd_lowres_intra_costs against golden results of the
reference's own Lowres::init steps (frameInitLowres + extendPicBorder) and LookaheadTLD::lowresIntraEstimate (tests/golden/lowres_golden.npz, generated
here from oracle/_ref by tests/golden/make_golden.py)."""
import hashlib
import os

import numpy as np
import pytest

import hevc_testlib as T

GOLD_PATH = os.path.join(T.GOLDEN_DIR, "lowres_golden.npz")
CASES = [(8, 11, (0, 0)), (8, 12, (8, 8)), (10, 13, (0, 0)), (10, 14, (24, 40))]        # (bit depth, seed, crop of the 256x192 scene)


def test_golden_is_varied_and_sums_follow_from_block_costs():
    g = np.load(GOLD_PATH)
    for k, (depth, seed, crop) in enumerate(CASES):
        c = T.lowres_case(depth, seed, crop)
        cost, mode = g["%d/cost" % k], g["%d/mode" % k]
        assert len(np.unique(mode)) > 8 and cost.min() > 9          # many modes win; the penalties are in
        lc, rows, est = T.lowres_frame_sums(c, cost)
        assert np.array_equal(lc, g["%d/lowres_costs" % k]) and np.array_equal(rows, g["%d/row_satds" % k]) and est == int(g["%d/sums" % k][0]) == int(g["%d/sums" % k][1])


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(len(CASES)))
def test_hip_lowres_matches_reference_golden(k):
    g = np.load(GOLD_PATH)
    depth, seed, crop = CASES[k]
    c = T.lowres_case(depth, seed, crop)
    planes, cost, mode = T.lowres_run_hip(T.load_hip(depth), c)
    for i, p in enumerate(planes):
        assert hashlib.md5(np.ascontiguousarray(p).tobytes()).hexdigest() == str(g["%d/plane_md5" % k][i]), "lowres plane %d" % i
    assert np.array_equal(cost, g["%d/cost" % k]), np.argwhere(cost != g["%d/cost" % k])[:5].tolist()
    assert np.array_equal(mode, g["%d/mode" % k]), np.argwhere(mode != g["%d/mode" % k])[:5].tolist()


COST_GOLD = os.path.join(T.GOLDEN_DIR, "lowres_cost_golden.npz")
# (bit depth, seed, crop, p0, b, p1): P candidates (p1 == b) and B candidates between two references
COST_CASES = [(8, 21, (0, 0), 0, 1, 1), (8, 21, (0, 0), 0, 2, 2), (8, 22, (0, 0), 0, 1, 2), (8, 23, (24, 8), 0, 1, 2), (10, 24, (0, 0), 0, 1, 1), (10, 25, (8, 40), 0, 1, 2)]


def test_cost_golden_is_varied():
    g = np.load(COST_GOLD)
    used = np.zeros(4, np.int64); nmv = 0
    for k in range(len(COST_CASES)):
        used += np.bincount(g["%d/lowres_costs" % k] >> 14, minlength=4)
        nmv += len(np.unique(g["%d/mvs" % k][0], axis=0))
    assert (used > 20).all() and nmv > 100, (used, nmv)        # intra, L0, L1 and bi-prediction all win somewhere; many different vectors


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(len(COST_CASES)))
def test_hip_lowres_frame_cost_matches_reference_golden(k):
    g = np.load(COST_GOLD)
    depth, seed, crop, p0, b, p1 = COST_CASES[k]
    c = T.lowres_cost_case(depth, seed, crop)
    got = T.lowres_cost_run_hip(T.load_hip(depth), T.HipME(depth), c, p0, b, p1)
    for name in ("intra_cost", "mvs", "mv_costs", "lowres_costs", "row_satds"):
        want = g["%d/%s" % (k, name)]
        assert np.array_equal(got[name], want), (name, np.argwhere(got[name] != want)[:6].tolist())
    want = g["%d/sums" % k]
    assert int(got["sums"][0]) == int(want[0]) and int(got["sums"][2]) == int(want[2]), (got["sums"], want)


@pytest.mark.gpu
def test_hip_lowres_frame_cost_many_estimates_in_one_batch_repeatedly():
    """The row chain's hand-over inside ONE launch (csrc/lowres_kernels.hip: lowres_cost_row -- a row's vectors and its progress word cross to the wavefront of the row above
    as agent-scope atomics behind a bare s_waitcnt, no cache-wide fence since round 5): sixty estimates of the golden cases' pictures in one batch, every one with buffers of its
    own, five batches in a row while the previous batch's buffers are still warm in the caches -- every estimate of every batch must be the reference's, bit for bit."""
    import ctypes as C
    import torch
    g = np.load(COST_GOLD)
    JOB = np.dtype([("d_fenc", "<u8"), ("d_ref0", "<u8", 4), ("d_ref1", "<u8", 4), ("d_intra_cost", "<u8"), ("d_mvs0", "<u8"), ("d_mv_costs0", "<u8"), ("d_mvs1", "<u8"),
                    ("d_mv_costs1", "<u8"), ("d_lowres_costs", "<u8"), ("d_bcost", "<u8"), ("do_search0", "<i4"), ("do_search1", "<i4"), ("rows_per_slice", "<i4"),
                    ("num_slices", "<i4"), ("d_ref0w", "<u8", 4)])
    for k in (0, 2):
        depth, seed, crop, p0, b, p1 = COST_CASES[k]
        c = T.lowres_cost_case(depth, seed, crop)
        L = T.load_hip(depth); me = T.HipME(depth)
        isz = c["lumas"][0].dtype.itemsize
        lw, lh = c["wcu"] * 8, c["hcu"] * 8
        lstride = (c["W"] // 2) + 2 * T.MC_MX
        lstride += (32 - (lstride & 31)) & 31
        rows = lh + 2 * T.MC_MY
        o = (T.MC_MY * lstride + T.MC_MX) * isz
        ncu = c["wcu"] * c["hcu"]
        planes, intra = [], []
        for f in range(3):
            d_src = torch.from_numpy(c["lumas"][f].view(np.uint8)).cuda()
            d_pl = [torch.zeros(rows * lstride * isz, dtype=torch.uint8, device="cuda") for _ in range(4)]
            ptrs = (C.c_void_p * 4)(*[q.data_ptr() + o for q in d_pl])
            assert L.lib.x265amd_lowres_init(None, C.c_void_p(d_src.data_ptr() + (T.MC_MY * c["stride"] + T.MC_MX) * isz), C.c_int64(c["stride"]), lw, lh, ptrs, C.c_int64(lstride),
                                             T.MC_MX, T.MC_MY) == 0
            d_cost = torch.zeros(ncu, dtype=torch.int32, device="cuda"); d_mode = torch.zeros(ncu, dtype=torch.uint8, device="cuda")
            assert L.lib.x265amd_lowres_intra_costs(None, C.c_void_p(d_pl[0].data_ptr() + o), C.c_int64(lstride), c["wcu"], c["hcu"], T.LOWRES_LAMBDA[depth],
                                                    C.c_void_p(d_cost.data_ptr()), C.c_void_p(d_mode.data_ptr())) == 0
            planes.append(d_pl); intra.append(d_cost)
        torch.cuda.synchronize()
        bidir = p1 > b
        N = 60
        for rep in range(5):
            jobs = np.zeros(N, JOB); bufs = []
            for i in range(N):
                mv = [torch.zeros(ncu * 2, dtype=torch.int16, device="cuda") for _ in range(2)]
                mc = [torch.zeros(ncu, dtype=torch.int32, device="cuda") for _ in range(2)]
                lc = torch.zeros(ncu, dtype=torch.int16, device="cuda"); bc = torch.zeros(ncu, dtype=torch.int32, device="cuda")
                bufs.append((mv, mc, lc, bc))
                j = jobs[i]
                j["d_fenc"] = planes[b][0].data_ptr() + o
                j["d_ref0"] = [q.data_ptr() + o for q in planes[p0]]
                if bidir:
                    j["d_ref1"] = [q.data_ptr() + o for q in planes[p1]]
                    j["d_mvs1"], j["d_mv_costs1"], j["do_search1"] = mv[1].data_ptr(), mc[1].data_ptr(), 1
                j["d_intra_cost"] = intra[b].data_ptr()
                j["d_mvs0"], j["d_mv_costs0"], j["do_search0"] = mv[0].data_ptr(), mc[0].data_ptr(), 1
                j["d_lowres_costs"], j["d_bcost"] = lc.data_ptr(), bc.data_ptr()
            assert L.lib.x265amd_lowres_frame_cost_batch(None, me.ctx, jobs.ctypes.data_as(C.c_void_p), N, C.c_int64(lstride), c["wcu"], c["hcu"]) == 0, L.lib.x265amd_last_error()
            torch.cuda.synchronize()
            for i, (mv, mc, lc, bc) in enumerate(bufs):
                got_lc = lc.cpu().numpy().view(np.uint16)
                assert np.array_equal(got_lc, g["%d/lowres_costs" % k]), (k, rep, i, np.argwhere(got_lc != g["%d/lowres_costs" % k])[:4].tolist())
                for l in range(2 if bidir else 1):
                    assert np.array_equal(mv[l].cpu().numpy().reshape(ncu, 2), g["%d/mvs" % k][l]), (k, rep, i, l)
                    assert np.array_equal(mc[l].cpu().numpy(), g["%d/mv_costs" % k][l]), (k, rep, i, l)


@pytest.mark.gpu
def test_hip_lowres_cost_sums_are_the_callers_sums():
    """x265amd_lowres_cost_sums against the reference golden's own sums (estimateFrameCost's score before the B scaling, and its intra block count) and against the
    caller's reduction written out in numpy, several estimates in one call, for a picture of two block rows as well"""
    import ctypes as C
    import torch
    g = np.load(COST_GOLD)
    L = T.load_hip(8)
    JOB = np.dtype([("d_fenc", "<u8"), ("d_ref0", "<u8", 4), ("d_ref1", "<u8", 4), ("d_intra_cost", "<u8"), ("d_mvs0", "<u8"), ("d_mv_costs0", "<u8"), ("d_mvs1", "<u8"),
                    ("d_mv_costs1", "<u8"), ("d_lowres_costs", "<u8"), ("d_bcost", "<u8"), ("do_search0", "<i4"), ("do_search1", "<i4"), ("rows_per_slice", "<i4"),
                    ("num_slices", "<i4"), ("d_ref0w", "<u8", 4)])
    for k in (0, 2, 3):
        depth, seed, crop, p0, b, p1 = COST_CASES[k]
        c = T.lowres_cost_case(depth, seed, crop)
        lc = g["%d/lowres_costs" % k]
        rng = np.random.default_rng(900 + k)
        variants = [(lc, rng.integers(0, 1 << 14, lc.size).astype(np.int32)) for _ in range(3)]
        variants.append((rng.integers(0, 1 << 16, lc.size).astype(np.uint16), rng.integers(0, 1 << 20, lc.size).astype(np.int32)))
        for (wcu, hcu) in ((c["wcu"], c["hcu"]), (c["wcu"], 2)):
            n = wcu * hcu
            jobs = np.zeros(len(variants), JOB); keep = []
            for i, (l, bc) in enumerate(variants):
                dl, db = torch.from_numpy(l[:n].view(np.int16).copy()).cuda(), torch.from_numpy(bc[:n].copy()).cuda()
                keep += [dl, db]
                jobs[i]["d_lowres_costs"], jobs[i]["d_bcost"] = dl.data_ptr(), db.data_ptr()
            sums = np.zeros(2 * len(variants), np.int64)
            assert L.lib.x265amd_lowres_cost_sums(None, jobs.ctypes.data_as(C.c_void_p), len(variants), wcu, hcu, sums.ctypes.data_as(C.c_void_p)) == 0
            for i, (l, bc) in enumerate(variants):
                cc = dict(c, wcu=wcu, hcu=hcu)
                score, est, intra_mbs, _ = T.lowres_cost_sums(cc, l[:n], bc[:n], False)
                assert (int(sums[2 * i]), int(sums[2 * i + 1])) == (est, intra_mbs), (k, wcu, hcu, i)


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [8, 10])
def test_hip_lowres_weight_costs_one_and_many(depth):
    """x265amd_lowres_weight_costs and its many-decisions form against the oracle's pieces in the reference's order (weightCostLuma, slicetype.cpp:826-858): the reference plane
    weighted by weight_pp_c, per 8x8 block min(satd_8x8(source, weighted reference), intra cost), uint32 sum -- unweighted, weighted, with and without the intra costs"""
    import ctypes as C
    import torch
    L, O = T.load_hip(depth), T.load_oracle(depth)
    dt = np.uint8 if depth == 8 else np.uint16
    pmax = (1 << depth) - 1
    rng = np.random.default_rng(7700 + depth)
    W, H, stride = 96, 56, 128
    CAND = np.dtype([("present", "<i4"), ("w0", "<i4"), ("round", "<i4"), ("shift", "<i4"), ("offset", "<i4")])
    JOB = np.dtype([("d_fenc", "<u8"), ("d_ref", "<u8", 4), ("d_mvs", "<u8"), ("d_intra_cost", "<u8"), ("cands", CAND, 2)])
    assert JOB.itemsize == 96

    def cand(scale, denom, off):
        corr = 14 - depth
        return (1, scale, (1 << (denom - 1) if denom else 0) << corr, denom + corr, off << (depth - 8))

    pairs, keep, want = [], [], []
    for k in range(5):
        base = np.clip(np.kron(rng.integers(0, pmax + 1, (H // 8 + 1, stride // 8)), np.ones((8, 8), np.int64)) + rng.integers(-20, 21, (H + 8, stride)) * (1 << (depth - 8)), 0, pmax)
        ref = base.astype(dt)
        fenc = np.clip(base * (0.6 + 0.1 * k) + 5 * k + rng.integers(-6, 7, base.shape), 0, pmax).astype(dt)
        intra = rng.integers(0, 3000, (H // 8) * (W // 8)).astype(np.int32) if k & 1 else None
        c2 = [(0, 0, 0, 0, 0), cand(int(rng.integers(30, 127)), int(rng.integers(0, 8)), int(rng.integers(-20, 21)))]
        exp = []
        for c in c2:
            wref = ref.copy()
            if c[0]:
                O.lib.orc_weight_pp(T._ptr(ref), T._ptr(wref), C.c_int64(stride), stride, H + 8, c[1], c[2], c[3], c[4])
            tot = 0
            for by in range(H // 8):
                for bx in range(W // 8):
                    v = O.call("satd", 1, wref[8 * by:8 * by + 8, 8 * bx:8 * bx + 8].copy(), 8, fenc[8 * by:8 * by + 8, 8 * bx:8 * bx + 8].copy(), 8)
                    tot += min(v, int(intra[by * (W // 8) + bx])) if intra is not None else v
            exp.append(tot & 0xFFFFFFFF)
        d_f, d_r = torch.from_numpy(fenc.view(np.uint8).copy()).cuda(), torch.from_numpy(ref.view(np.uint8).copy()).cuda()
        d_i = torch.from_numpy(intra).cuda() if intra is not None else None
        keep += [d_f, d_r, d_i]
        pairs.append((d_f, d_r, d_i, c2)); want.append(exp)
        # one decision
        cands = np.array(c2, CAND); costs = np.zeros(2, np.uint32)
        refs = (C.c_void_p * 4)(d_r.data_ptr(), None, None, None)
        assert L.lib.x265amd_lowres_weight_costs(None, C.c_void_p(d_f.data_ptr()), refs, None, C.c_void_p(d_i.data_ptr()) if d_i is not None else None, C.c_int64(stride), W, H,
                                                 cands.ctypes.data_as(C.c_void_p), 2, costs.ctypes.data_as(C.c_void_p)) == 0, L.lib.x265amd_last_error()
        assert [int(v) for v in costs] == exp, (k, costs, exp)
    # the five decisions at once
    jobs = np.zeros(len(pairs), JOB)
    for i, (d_f, d_r, d_i, c2) in enumerate(pairs):
        jobs[i]["d_fenc"], jobs[i]["d_ref"][0], jobs[i]["d_intra_cost"] = d_f.data_ptr(), d_r.data_ptr(), d_i.data_ptr() if d_i is not None else 0
        jobs[i]["cands"] = np.array(c2, CAND)
    costs = np.zeros(2 * len(pairs), np.uint32)
    assert L.lib.x265amd_lowres_weight_costs_many(None, jobs.ctypes.data_as(C.c_void_p), len(pairs), C.c_int64(stride), W, H, costs.ctypes.data_as(C.c_void_p)) == 0, L.lib.x265amd_last_error()
    assert [int(v) for v in costs] == [v for e in want for v in e]


AQ_GOLD = os.path.join(T.GOLDEN_DIR, "aq_energy_golden.npz")
AQ_CASES = [(8, 41, T.MC_W, T.MC_H, 16), (8, 42, T.MC_W - 8, T.MC_H - 24, 16), (8, 43, T.MC_W, T.MC_H, 8), (10, 44, T.MC_W - 40, T.MC_H, 16), (10, 45, T.MC_W, T.MC_H - 8, 8)]


def test_aq_golden_shapes():
    g = np.load(AQ_GOLD)
    for k, (depth, seed, W, H, qg) in enumerate(AQ_CASES):
        assert len(g["%d/energy" % k]) == ((W + qg - 1) // qg) * ((H + qg - 1) // qg) and g["%d/energy" % k].max() > 1000 and g["%d/wp" % k].min() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(len(AQ_CASES)))
def test_hip_aq_energy_matches_reference_golden(k):
    g = np.load(AQ_GOLD)
    depth, seed, W, H, qg = AQ_CASES[k]
    energy, wp = T.aq_run_hip(T.load_hip(depth), T.aq_case(depth, seed), W, H, qg)
    assert np.array_equal(energy, g["%d/energy" % k]), np.argwhere(energy != g["%d/energy" % k])[:5].tolist()
    assert np.array_equal(wp, g["%d/wp" % k]), (wp, g["%d/wp" % k])


# adaptive quantisation, host half: (AQ_CASES index, aq-mode, strength, bias strength)
AQ_OFFSET_CASES = [(0, 1, 1.0, 1.0), (0, 2, 1.0, 1.0), (0, 3, 0.8, 1.5), (2, 2, 1.2, 1.0), (3, 1, 0.6, 1.0), (3, 3, 1.0, 1.0), (4, 2, 1.0, 1.0), (1, 2, 1.0, 1.0)]


@pytest.mark.parametrize("k", range(len(AQ_OFFSET_CASES)))
def test_aq_offsets_match_reference_golden_bit_for_bit(k):
    """x265amd_aq_offsets (host code of the library, no GPU work) on the golden block energies against the reference's calcAdaptiveQuantFrame: the doubles
    must be identical, not close"""
    g = np.load(AQ_GOLD)
    ci, mode, strength, bias = AQ_OFFSET_CASES[k]
    depth, seed, W, H, qg = AQ_CASES[ci]
    energy = g["%d/energy" % ci]
    avg = (((W // 2) + 7) >> 3) * (((H // 2) + 7) >> 3) * (4 if qg == 8 else 1)
    a, t, f = T.aq_offsets_prod(T.load_hip(depth), energy, avg, mode, strength, bias, qg)
    n = len(energy)
    assert np.array_equal(a.view(np.uint64), g["o%d/qp_aq_offset" % k][:n].view(np.uint64)), float(np.abs(a - g["o%d/qp_aq_offset" % k][:n]).max())
    assert np.array_equal(t.view(np.uint64), g["o%d/qp_cutree_offset" % k][:n].view(np.uint64))
    assert np.array_equal(f, g["o%d/inv_qscale_factor" % k][:n])
