"""Reference-plane production: extendPicBorder and MotionReference::applyWeight.  Oracle against the
reference's own function / class (oracle/_ref) and a committed digest; the GPU kernels against the oracle."""
import hashlib
import os

import numpy as np
import pytest

import hevc_testlib as T

GOLD = {8: "d0", 10: "d1"}


def digest(res):
    h = hashlib.sha256()
    for a, b in res:
        h.update(a.tobytes()); h.update(b.tobytes())
    return h.hexdigest()


def load_gold():
    import json
    with open(os.path.join(T.GOLDEN_DIR, "planes_golden.json")) as f:
        return json.load(f)


@pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("depth", [8, 10])
def test_oracle_vs_reference(depth):
    R, O = T.load_ref(depth), T.load_oracle(depth)
    for seed in (1, 2):
        cases = T.plane_cases(depth, seed)
        for i, ((a, b), (x, y)) in enumerate(zip(T.plane_run_host(R, cases), T.plane_run_host(O, cases))):
            assert np.array_equal(a, x), (i, "extend")
            assert np.array_equal(b, y), (i, "weight")


@pytest.mark.parametrize("depth", [8, 10])
def test_oracle_matches_golden(depth):
    assert digest(T.plane_run_host(T.load_oracle(depth), T.plane_cases(depth, 1))) == load_gold()[str(depth)]


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [8, 10])
def test_hip_planes(depth):
    H, O = T.load_hip(depth), T.load_oracle(depth)
    for seed in (1, 2, 3):
        cases = T.plane_cases(depth, seed)
        got, want = T.plane_run_hip(H, cases), T.plane_run_host(O, cases)
        for i, ((a, b), (x, y)) in enumerate(zip(got, want)):
            assert np.array_equal(a, x), (i, "extend")
            assert np.array_equal(b, y), (i, "weight")
        if seed == 1:
            assert digest(got) == load_gold()[str(depth)]
