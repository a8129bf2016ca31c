"""Pins the inter-prediction oracle (oracle/hevc_oracle_mc.c) against the reference's own Predict::motionCompensation
(uni / bi / weighted, luma + 4:2:0 chroma, clipMv) driven through oracle/_ref/librefprims*.so.  This container only."""
import numpy as np
import pytest

import hevc_testlib as T

pytestmark = pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref not built (needs /root/reference)")


@pytest.mark.parametrize("depth", [8, 10])
def test_motion_compensation(depth):
    ref, orc = T.load_ref(depth), T.load_oracle(depth)
    for seed in range(3):
        pics, stride, cstride, org = T.mc_make_refs(depth, 900 + seed)
        jobs = T.mc_jobs(900 + seed, 400)
        want = T.mc_run_host(ref, pics, stride, cstride, org, jobs)
        got = T.mc_run_host(orc, pics, stride, cstride, org, jobs)
        for i, (w, g) in enumerate(zip(want, got)):
            for c in range(3):
                if w[c] is None:
                    assert g[c] is None
                else:
                    assert np.array_equal(w[c], g[c]), (i, c, jobs[i])
