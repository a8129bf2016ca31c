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
