"""Pins the motion-estimation oracle (oracle/hevc_oracle_me.c) against the reference's own BitCost and
MotionEstimate classes driven through oracle/_ref/librefprims*.so.  This container only."""
import ctypes as C

import numpy as np
import pytest

import hevc_testlib as T

pytestmark = pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref not built (needs /root/reference)")


@pytest.mark.parametrize("depth", [8, 10])
def test_lambda_and_mvcost_tables(depth):
    ref, orc = T.load_ref(depth), T.load_oracle(depth)
    ref.lib.ref_tbl_lambda.restype = C.POINTER(C.c_double)
    orc.lib.orc_lambda.restype = C.c_double
    lam = ref.lib.ref_tbl_lambda()
    for qp in range(70):
        assert orc.lib.orc_lambda(qp) == lam[qp], qp
    ref.lib.ref_mvcost_table.restype = C.POINTER(C.c_uint16)
    orc.lib.orc_mvcost_table.restype = C.POINTER(C.c_uint16)
    n = 2 * 65536 + 1
    for qp in list(range(0, 70, 3)) + [51, 69]:
        a = np.ctypeslib.as_array(C.cast(C.addressof(ref.lib.ref_mvcost_table(qp).contents) - 2 * 65536, C.POINTER(C.c_uint16)), (n,))
        b = np.ctypeslib.as_array(C.cast(C.addressof(orc.lib.orc_mvcost_table(qp).contents) - 2 * 65536, C.POINTER(C.c_uint16)), (n,))
        assert np.array_equal(a, b), qp


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("method,subme", [(T.ME_HEX, 2), (T.ME_HEX, 0), (T.ME_HEX, 1), (T.ME_HEX, 5), (T.ME_HEX, 7),
                                          (T.ME_DIA, 0), (T.ME_DIA, 2), (T.ME_STAR, 2), (T.ME_STAR, 4)])
def test_motion_estimate(depth, method, subme):
    ref, orc = T.load_ref(depth), T.load_oracle(depth)
    for seed, motion in ((1, (5, -3)), (2, (-17, 9)), (3, (0, 0)), (4, (33, 21))):
        cur, rp, stride, origin = T.me_make_planes(depth, seed, motion=motion)
        jobs = T.me_jobs(seed * 100 + method * 10 + subme, 60, motion=motion, methods=(method,), submes=(subme,))
        want = T.me_run_host(ref, cur, rp, stride, origin, jobs)
        got = T.me_run_host(orc, cur, rp, stride, origin, jobs)
        bad = np.argwhere((want != got).any(axis=1))
        assert len(bad) == 0, "job %d: %s want %s got %s" % (bad[0][0], jobs[int(bad[0][0])], want[int(bad[0][0])], got[int(bad[0][0])])
        assert len({tuple(r[:2]) for r in want}) > 3      # the searches actually move


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("method,subme", [(T.ME_STAR, 3), (T.ME_HEX, 3), (T.ME_STAR, 4), (T.ME_HEX, 7), (T.ME_DIA, 5)])
def test_motion_estimate_chroma_satd(depth, method, subme):
    """encoder form of setSourcePU: chroma SATD joins every subpelCompare when subme > 2 (motion.cpp:234-237, :1625-1686)"""
    ref, orc = T.load_ref(depth), T.load_oracle(depth)
    for seed, motion in ((11, (6, -4)), (12, (-18, 10))):
        cur, rp, stride, cstride, origin, corg = T.me_make_yuv(depth, seed, motion=motion)
        jobs = T.me_jobs(seed * 100 + method * 10 + subme, 50, motion=motion, methods=(method,), submes=(subme,))
        want = T.me_run_host_c(ref, cur, rp, stride, cstride, origin, corg, jobs)
        got = T.me_run_host_c(orc, cur, rp, stride, cstride, origin, corg, jobs)
        bad = np.argwhere((want != got).any(axis=1))
        assert len(bad) == 0, "job %d: %s want %s got %s" % (bad[0][0], jobs[int(bad[0][0])], want[int(bad[0][0])], got[int(bad[0][0])])
        # chroma really contributes: the luma-only search gives different costs
        luma_only = T.me_run_host(ref, cur[0], rp[0], stride, origin, jobs)
        assert (luma_only[:, 2] != want[:, 2]).sum() > 10
