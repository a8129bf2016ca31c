"""Distortion of inter prediction candidates (selectMVP / mergeEstimation / merge scan / bi-prediction tries): the oracle
against the reference's own Predict + primitives (oracle/_ref) and against committed golden costs; the GPU entry
x265amd_inter_cost against the oracle."""
import os

import numpy as np
import pytest

import hevc_testlib as T

GOLD_PATH = os.path.join(T.GOLDEN_DIR, "inter_cost_golden.npz")


def scene(depth, seed):
    pics, stride, cstride, org = T.mc_make_refs(depth, 1300 + seed, nref=4)
    return pics[:3], pics[3], stride, cstride, org


@pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("depth", [8, 10])
def test_oracle_vs_reference(depth):
    R, O = T.load_ref(depth), T.load_oracle(depth)
    for seed in range(2):
        pics, fenc, stride, cstride, org = scene(depth, seed)
        jobs = T.inter_cost_jobs(40 + seed, 500)
        a, pa = T.inter_cost_run_host(R, pics, fenc, stride, cstride, org, jobs)
        b, pb = T.inter_cost_run_host(O, pics, fenc, stride, cstride, org, jobs)
        assert np.array_equal(a, b), np.nonzero((a != b).any(1))[0][:10]
        assert all(np.array_equal(x, y) for x, y in zip(pa, pb))
        assert len(np.unique(a[:, 0])) > 200 and (a[:, 1] > 0).sum() > 50


@pytest.mark.parametrize("depth", [8, 10])
def test_oracle_matches_golden(depth):
    gold = np.load(GOLD_PATH)
    O = T.load_oracle(depth)
    for seed in range(2):
        pics, fenc, stride, cstride, org = scene(depth, seed)
        cost, _ = T.inter_cost_run_host(O, pics, fenc, stride, cstride, org, T.inter_cost_jobs(40 + seed, 500))
        assert np.array_equal(cost, gold["cost/%d/%d" % (depth, seed)])


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [8, 10])
def test_hip_inter_cost(depth):
    gold = np.load(GOLD_PATH)
    H, O = T.load_hip(depth), T.load_oracle(depth)
    for seed in range(3):
        pics, fenc, stride, cstride, org = scene(depth, seed)
        jobs = T.inter_cost_jobs(40 + seed, 500)
        a, pa = T.inter_cost_run_hip(H, pics, fenc, stride, cstride, org, jobs)
        b, pb = T.inter_cost_run_host(O, pics, fenc, stride, cstride, org, jobs)
        assert np.array_equal(a, b), np.nonzero((a != b).any(1))[0][:10]
        assert all(np.array_equal(x, y) for x, y in zip(pa, pb))
        if seed < 2:
            assert np.array_equal(a, gold["cost/%d/%d" % (depth, seed)])
