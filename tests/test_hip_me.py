"""GPU parity for the fused motion-estimation kernel (x265amd_me_search) against the CPU oracle and the golden
results produced by the reference's own MotionEstimate class; bit-exact (integer path)."""
import hashlib
import os

import numpy as np
import pytest

import hevc_testlib as T

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(T.GOLDEN_DIR, "me_golden.npz"))
CONFIGS = sorted({tuple(int(v) for v in k.split("/")[2:4]) for k in GOLD.files if k.startswith("me/")})
SCENES = ((1, (5, -3)), (2, (-17, 9)), (3, (0, 0)), (4, (33, 21)))


@pytest.mark.parametrize("depth", [8, 10])
def test_mvcost_tables(depth):
    me = T.HipME(depth)
    want = GOLD["mvcost_sha256/%d" % depth]
    for qp in range(70):
        assert hashlib.sha256(me.host_mvcost(qp).tobytes()).digest() == want[qp].tobytes(), qp
    me.close()


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("method,subme", CONFIGS)
def test_me_search(depth, method, subme):
    me, orc = T.HipME(depth), T.load_oracle(depth)
    for seed, motion in SCENES:
        cur, rp, stride, origin = T.me_make_planes(depth, seed, motion=motion)
        jobs = T.me_jobs(seed * 100 + method * 10 + subme, 60, motion=motion, methods=(method,), submes=(subme,))
        want = T.me_run_host(orc, cur, rp, stride, origin, jobs)
        assert np.array_equal(want, GOLD["me/%d/%d/%d/%d" % (depth, method, subme, seed)])
        got = me.run(cur, rp, stride, origin, jobs)
        bad = np.argwhere((want != got).any(axis=1))
        assert len(bad) == 0, "job %d: %s want %s got %s" % (bad[0][0], jobs[int(bad[0][0])], want[int(bad[0][0])], got[int(bad[0][0])])
    me.close()


@pytest.mark.parametrize("depth", [8, 10])
def test_me_small_window_fallback(depth):
    """a window far smaller than the search areas forces the direct-from-HBM path; results must not change"""
    me, orc = T.HipME(depth), T.load_oracle(depth)
    cur, rp, stride, origin = T.me_make_planes(depth, 7, motion=(9, 6))
    jobs = T.me_jobs(77, 80, motion=(9, 6), methods=(T.ME_HEX, T.ME_STAR, T.ME_DIA), submes=(2, 3))
    want = T.me_run_host(orc, cur, rp, stride, origin, jobs)
    for win in ((80, 80), (128, 96), (256, 200)):
        got = me.run(cur, rp, stride, origin, jobs, max_win=win)
        assert np.array_equal(want, got), win
    me.close()


CHROMA_CONFIGS = sorted({tuple(int(v) for v in k.split("/")[2:4]) for k in GOLD.files if k.startswith("mec/")})


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("method,subme", CHROMA_CONFIGS)
def test_me_search_chroma_satd(depth, method, subme):
    """encoder form (setSourcePU from a CU Yuv with bChroma): chroma SATD joins the sub-pel comparisons when subme > 2"""
    me = T.HipME(depth)
    for seed, motion in ((11, (6, -4)), (12, (-18, 10))):
        cur, rp, stride, cstride, origin, corg = T.me_make_yuv(depth, seed, motion=motion)
        jobs = T.me_jobs(seed * 100 + method * 10 + subme, 50, motion=motion, methods=(method,), submes=(subme,))
        want = GOLD["mec/%d/%d/%d/%d" % (depth, method, subme, seed)]
        got = me.run_c(cur, rp, stride, cstride, origin, corg, jobs)
        bad = np.argwhere((want != got).any(axis=1))
        assert len(bad) == 0, "job %d: %s want %s got %s" % (bad[0][0], jobs[int(bad[0][0])], want[int(bad[0][0])], got[int(bad[0][0])])
    me.close()
