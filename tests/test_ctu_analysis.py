"""CTU mode decision of inter slices (x265amd_compress_ctu_inter) against golden results of the reference's own
Analysis::compressCTU run on CUData / Slice / Frame / MotionReference fixtures of the same pictures and maps
(tests/golden/ctu_analysis_golden.npz, generated here from oracle/_ref by tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

import hevc_testlib as T

# (depth, seed, early skip, rskip, psy-rd, tuQTMaxInterDepth, limit-refs[, B slice, b-intra[, I slice]])
CASES = [(8, 1, 1, 1, 2.0, 1, 0), (8, 2, 0, 1, 2.0, 2, 0), (8, 3, 1, 0, 0.0, 1, 0), (10, 4, 0, 1, 2.0, 1, 0), (8, 5, 1, 1, 1.0, 3, 0), (8, 6, 0, 0, 2.0, 1, 0),
         (8, 7, 1, 1, 2.0, 1, 3), (8, 8, 0, 0, 2.0, 1, 3), (10, 9, 0, 1, 0.0, 2, 1), (8, 10, 0, 1, 2.0, 1, 2),
         (8, 11, 1, 1, 2.0, 1, 3, 0, 0), (8, 12, 0, 1, 2.0, 1, 0, 0, 0), (10, 13, 0, 0, 0.0, 2, 3, 0, 0), (8, 14, 0, 1, 2.0, 1, 3, 1, 1), (8, 15, 1, 0, 1.0, 1, 0, 1, 1),
         (8, 16, 0, 0, 2.0, 1, 0, 0, 0, 1), (10, 17, 0, 0, 0.0, 1, 0, 0, 0, 1), (8, 18, 0, 0, 1.0, 1, 0, 0, 0, 1)]
# rectangular / asymmetric partitions and rd 5-6: (depth, seed, early skip, rskip, psy-rd, limit-refs, B slice, b-intra, rect, amp, limit-modes[, rd level])
PART_CASES = [(8, 21, 0, 1, 2.0, 3, 1, 0, 1, 0, 0), (8, 22, 0, 0, 2.0, 0, 0, 0, 1, 1, 0), (10, 23, 1, 1, 0.0, 3, 1, 1, 1, 1, 1), (8, 24, 0, 1, 1.0, 1, 0, 0, 1, 1, 1),
              (8, 25, 0, 0, 2.0, 2, 1, 0, 1, 1, 0), (8, 26, 0, 1, 2.0, 3, 0, 0, 1, 0, 1),
              (8, 31, 0, 1, 2.0, 3, 1, 1, 0, 0, 0, 5), (8, 32, 1, 0, 2.0, 0, 0, 0, 1, 1, 0, 6), (10, 33, 0, 1, 0.0, 1, 1, 0, 1, 1, 1, 5), (8, 34, 1, 1, 1.0, 3, 0, 0, 1, 0, 1, 6),
              (8, 35, 0, 0, 2.0, 2, 1, 1, 1, 1, 0, 5),
              (8, 41, 1, 1, 0.0, 3, 1, 1, 0, 0, 0, 2), (8, 42, 0, 1, 2.0, 0, 0, 0, 1, 1, 1, 2), (10, 43, 0, 0, 0.0, 1, 1, 0, 1, 0, 0, 2), (8, 44, 1, 0, 1.0, 2, 0, 0, 0, 0, 0, 2)]
# RDOQ / fast-intra: (depth, seed, early skip, rskip, psy-rd, tu-inter-depth, limit-refs, B slice, b-intra, rect, amp, limit-modes, rd level, rdoq level, psy-rdoq * 256, fast intra)
RDOQ_CASES = [(8, 51, 0, 1, 2.0, 3, 0, 1, 1, 1, 1, 0, 6, 2, 256, 0), (8, 52, 0, 0, 2.0, 3, 3, 0, 0, 1, 1, 1, 5, 2, 256, 0), (10, 53, 1, 1, 2.0, 2, 1, 1, 0, 0, 0, 0, 3, 1, 0, 0),
              (8, 54, 0, 1, 0.0, 1, 3, 1, 1, 1, 0, 1, 4, 2, 640, 0), (8, 55, 1, 1, 2.0, 1, 3, 1, 0, 0, 0, 0, 2, 0, 0, 1), (8, 56, 0, 1, 2.0, 3, 0, 0, 0, 1, 1, 0, 6, 2, 256, 0)]
GOLD_PATH = os.path.join(T.GOLDEN_DIR, "ctu_analysis_golden.npz")


def make_case(k):
    if k >= len(CASES) + len(PART_CASES):
        depth, seed, es, rs, psy, td, lr, is_b, b_intra, rect, amp, lm, rd, rq, prq, fi = RDOQ_CASES[k - len(CASES) - len(PART_CASES)]
        return T.ctu_case(depth, seed, is_b=bool(is_b), early_skip=es, rskip=rs, psy_rd=psy, tu_inter_depth=td, limit_refs=lr, b_intra=b_intra, rect=rect, amp=amp, limit_modes=lm,
                          rd_level=rd, rdoq_level=rq, psy_rdoq_scale=prq, fast_intra=fi)
    if k >= len(CASES):
        depth, seed, es, rs, psy, lr, is_b, b_intra, rect, amp, lm = PART_CASES[k - len(CASES)][:11]
        rd = PART_CASES[k - len(CASES)][11] if len(PART_CASES[k - len(CASES)]) > 11 else 3
        return T.ctu_case(depth, seed, is_b=bool(is_b), early_skip=es, rskip=rs, psy_rd=psy, limit_refs=lr, b_intra=b_intra, rect=rect, amp=amp, limit_modes=lm, rd_level=rd)
    depth, seed, es, rs, psy, td, lr = CASES[k][:7]
    is_b, b_intra = (CASES[k][7], CASES[k][8]) if len(CASES[k]) > 7 else (1, 0)
    return T.ctu_case(depth, seed, is_b=bool(is_b), early_skip=es, rskip=rs, psy_rd=psy, tu_inter_depth=td, limit_refs=lr, b_intra=b_intra,
                      intra_slice=len(CASES[k]) > 9 and bool(CASES[k][9]))


def test_golden_outcomes_are_varied():
    gold = np.load(GOLD_PATH)
    depths = np.zeros(4, np.int64); modes = np.zeros(4, np.int64); coded = 0
    parts = np.zeros(8, np.int64)
    for k in range(len(CASES) + len(PART_CASES) + len(RDOQ_CASES)):
        for i in range(3):
            u = gold["%d/%d/units" % (k, i)]
            parts += np.bincount(u[:, 2][u[:, 1] == T.MODE_INTER], minlength=8)
            depths += np.bincount(u[:, 0], minlength=4); modes += np.bincount(u[:, 1], minlength=4); coded += int((u[:, 4:7] > 0).any(1).sum())
    assert (depths[1:] > 100).all() and modes[1] > 500 and modes[3] > 500 and coded > 300, (depths, modes, coded)
    assert (parts[[1, 2]] > 50).all() and (parts[4:] > 0).all(), parts          # every rectangular / asymmetric partition was chosen somewhere


@pytest.mark.gpu
def test_hip_compress_ctu_inter_matches_reference_golden():
    gold = np.load(GOLD_PATH)
    mes = {}
    for k, cfg in enumerate(CASES + PART_CASES + RDOQ_CASES):
        depth = cfg[0]
        if depth not in mes:
            mes[depth] = T.HipME(depth)
        c = make_case(k)
        got = T.ctu_pack(T.ctu_run_hip(T.load_hip(depth), mes[depth], c))
        for i, d in enumerate(got):
            for name, a in d.items():
                want = gold["%d/%d/%s" % (k, i, name)]
                if not np.array_equal(a, want):
                    bad = np.argwhere(np.asarray(a) != want)[:6].tolist()
                    raise AssertionError("case %d CTU %d (addr %d): %s differs from the reference's result at %s: got %s want %s" % (
                        k, i, c["ctus"][i], name, bad, np.asarray(a)[tuple(np.array(bad).T)].tolist(), want[tuple(np.array(bad).T)].tolist()))
