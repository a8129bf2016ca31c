"""Inter prediction against golden digests of the reference's own Predict::motionCompensation (tests/golden/mc_golden.npz):
the oracle on CPU, the fused HIP kernel on the GPU (which is also compared sample by sample with the oracle)."""
import os

import numpy as np
import pytest

import hevc_testlib as T

GOLD = np.load(os.path.join(T.GOLDEN_DIR, "mc_golden.npz"))


@pytest.mark.parametrize("depth", [8, 10])
def test_oracle(depth):
    orc = T.load_oracle(depth)
    for seed in range(3):
        pics, stride, cstride, org = T.mc_make_refs(depth, 900 + seed)
        res = T.mc_run_host(orc, pics, stride, cstride, org, T.mc_jobs(900 + seed, 400))
        assert T.mc_digest(res) == GOLD["mc/%d/%d" % (depth, seed)].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [8, 10])
def test_hip_motion_compensation(depth):
    hip, orc = T.load_hip(depth), T.load_oracle(depth)
    for seed in range(3):
        pics, stride, cstride, org = T.mc_make_refs(depth, 900 + seed)
        jobs = T.mc_jobs(900 + seed, 400)
        want = T.mc_run_host(orc, pics, stride, cstride, org, jobs)
        got = T.mc_run_hip(hip, pics, stride, cstride, org, jobs)
        for i, (w, g) in enumerate(zip(want, got)):
            for c in range(3):
                assert (w[c] is None) == (g[c] is None)
                if w[c] is not None:
                    assert np.array_equal(w[c], g[c]), (i, c, jobs[i])
        assert T.mc_digest(got) == GOLD["mc/%d/%d" % (depth, seed)].tobytes()
