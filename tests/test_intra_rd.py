"""Intra candidates of inter-slice CUs (x265amd_intra_in_inter; SURVEY row a7, RD side) against golden results of the reference's own
Search::checkIntraInInter + encodeIntraInInter run on CUData / Slice / Search / PicYuv fixtures (tests/golden/intra_rd_golden.npz, generated
here from oracle/_ref by tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

import hevc_testlib as T

# (depth, seed, slice type (0 B, 1 P), psy-rd, strong intra smoothing)
CASES = [(8, 601, 1, 2.0, 1), (8, 602, 0, 0.0, 1), (10, 603, 1, 2.0, 1), (8, 604, 1, 1.0, 0), (10, 605, 0, 0.0, 0), (8, 606, 1, 2.0, 1)]
GOLD_PATH = os.path.join(T.GOLDEN_DIR, "intra_rd_golden.npz")


def test_golden_covers_modes_and_edges():
    gold = np.load(GOLD_PATH)
    modes = set(); cmodes = set(); nocbf = 0
    for k in range(len(CASES)):
        for i in range(10):
            d = gold["%d/%d/dirs" % (k, i)]
            modes.add(int(d[0, 0])); cmodes.add(int(d[0, 1]))
            nocbf += int(not gold["%d/%d/units" % (k, i)][:, 1:4].any())
    assert len(modes) >= 8 and len(cmodes) >= 4, (modes, cmodes)


@pytest.mark.gpu
def test_hip_intra_in_inter_matches_reference_golden():
    gold = np.load(GOLD_PATH)
    for k, (depth, seed, st, psy, strong) in enumerate(CASES):
        c = T.intra_rd_case(depth, seed, st, psy, strong=strong)
        got = T.intra_rd_pack(T.intra_rd_run_hip(T.load_hip(depth), c), c)
        for i, d in enumerate(got):
            for name in ("info", "dirs", "pred", "units", "coeff", "recon", "res", "ctx"):
                want = gold["%d/%d/%s" % (k, i, name)]
                a = np.asarray(d[name])
                if not np.array_equal(a, want):
                    bad = np.argwhere(a != want)[:6].tolist() if a.shape == want.shape else "shape"
                    raise AssertionError("case %d CU %d (x %d y %d log2 %d qp %d): %s differs from the reference's result at %s: got %s want %s" % (
                        k, i, c["cus"][i]["x"], c["cus"][i]["y"], c["cus"][i]["log2_size"], c["cus"][i]["qp"], name, bad,
                        a[tuple(np.array(bad).T)].tolist() if bad != "shape" else a.shape, want[tuple(np.array(bad).T)].tolist() if bad != "shape" else want.shape))


# ---- Search::checkIntra: (depth, seed, slice type (2 I, 1 P, 0 B), psy-rd, strong intra smoothing) ----
CHECK_CASES = [(8, 701, 2, 2.0, 1), (8, 702, 2, 0.0, 1), (10, 703, 2, 2.0, 0), (8, 704, 1, 2.0, 1), (10, 705, 2, 1.0, 1), (8, 706, 2, 2.0, 1),
               (8, 707, 2, 2.0, 1, 4), (8, 708, 1, 0.0, 1, 4), (10, 709, 2, 2.0, 1, 4),           # 6th field: tu-intra-depth 4 (three levels of transform splits)
               (8, 711, 1, 2.0, 1), (8, 722, 0, 2.0, 1)]                                           # P / B slices without delta QP (tests/test_intra_cu_bits.py)
CHECK_GOLD_PATH = os.path.join(T.GOLDEN_DIR, "check_intra_golden.npz")


def test_check_intra_golden_covers_nxn_and_splits():
    gold = np.load(CHECK_GOLD_PATH)
    nxn = split = 0
    for k in range(len(CHECK_CASES)):
        for i in range(10):
            nxn += int(gold["%d/%d/dirs" % (k, i)][0, 2] == 3)
            split += int(gold["%d/%d/units" % (k, i)][:, 0].max() > int(gold["%d/%d/dirs" % (k, i)][0, 2] == 3))
    assert nxn >= 5 and split >= 3, (nxn, split)


@pytest.mark.gpu
def test_hip_check_intra_matches_reference_golden():
    gold = np.load(CHECK_GOLD_PATH)
    for k, cfg in enumerate(CHECK_CASES):
        depth, seed, st, psy, strong = cfg[:5]
        c = T.check_intra_case(depth, seed, st, psy, strong=strong, tu_intra=cfg[5] if len(cfg) > 5 else 0)
        got = T.intra_rd_pack(T.check_intra_run_hip(T.load_hip(depth), c), c)
        for i, d in enumerate(got):
            for name in ("dirs", "pred", "units", "coeff", "recon", "res", "ctx"):
                want = gold["%d/%d/%s" % (k, i, name)]
                a = np.asarray(d[name])
                if not np.array_equal(a, want):
                    bad = np.argwhere(a != want)[:6].tolist() if a.shape == want.shape else "shape"
                    raise AssertionError("case %d CU %d (x %d y %d log2 %d qp %d part %d): %s differs from the reference's result at %s: got %s want %s" % (
                        k, i, c["cus"][i]["x"], c["cus"][i]["y"], c["cus"][i]["log2_size"], c["cus"][i]["qp"], c["parts"][i], name, bad,
                        a[tuple(np.array(bad).T)].tolist() if bad != "shape" else a.shape, want[tuple(np.array(bad).T)].tolist() if bad != "shape" else want.shape))
