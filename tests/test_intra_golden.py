"""Intra neighbour set + 35-mode scan against golden results of the reference's Predict class and primitives
(tests/golden/intra_golden.npz): the oracle on CPU, the fused HIP kernel on the GPU."""
import os

import numpy as np
import pytest

import hevc_testlib as T

GOLD = np.load(os.path.join(T.GOLDEN_DIR, "intra_golden.npz"))


def check(run, L, depth):
    for seed in range(4):
        res = run(L, T.intra_cases(depth, 500 + seed, 200))
        assert np.array_equal(np.concatenate([r[0] for r in res]), GOLD["intra/%d/%d/ref" % (depth, seed)])
        assert np.array_equal(np.concatenate([r[1] for r in res if r[1] is not None]), GOLD["intra/%d/%d/flt" % (depth, seed)])
        assert np.array_equal(np.stack([r[2] for r in res]), GOLD["intra/%d/%d/sa8d" % (depth, seed)])


@pytest.mark.parametrize("depth", [8, 10])
def test_oracle(depth):
    check(T.intra_run_host, T.load_oracle(depth), depth)


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [8, 10])
def test_hip_intra_scan(depth):
    check(T.intra_run_hip, T.load_hip(depth), depth)
