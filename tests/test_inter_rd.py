"""Residual RD of inter CUs (x265amd_inter_residual_rd) against the reference's own Search::encodeResAndCalcRdInterCU.

CPU (not gpu): the product's host stages -- planning, the decision walk with the bit-counting CABAC coder, the final cost -- are driven with
the oracle executing the transform-chain records, and compared (a) with the reference driver when oracle/_ref is present, (b) with the
committed golden results (tests/golden/inter_rd_golden.npz, generated here from oracle/_ref by tests/golden/make_golden.py).
GPU: the whole entry point (both launches + the walk) against the same golden results."""
import os

import numpy as np
import pytest

import hevc_testlib as T

# (depth, seed, slice type (0 B, 1 P), tuQTMaxInterDepth, psy-rd)
CASES = [(8, 201, 1, 1, 2.0), (8, 202, 0, 2, 2.0), (8, 203, 1, 3, 0.0), (8, 204, 0, 4, 1.0), (10, 205, 1, 2, 2.0), (10, 206, 0, 3, 0.0), (8, 207, 1, 3, 2.0), (8, 208, 0, 1, 0.0)]
GOLD_PATH = os.path.join(T.GOLDEN_DIR, "inter_rd_golden.npz")


def check_golden(res, c, gold, k):
    packed = T.rd_pack(res, c)
    for i, d in enumerate(packed):
        for name, a in d.items():
            want = gold["%d/%d/%s" % (k, i, name)]
            assert np.array_equal(a, want), "case %d CU %d (log2 %d qp %d): %s differs from the reference's result" % (
                k, i, c["cus"][i]["log2_size"], c["cus"][i]["qp"], name)


def test_host_stages_match_golden():
    gold = np.load(GOLD_PATH)
    for k, (depth, seed, st, td, psy) in enumerate(CASES):
        c = T.rd_case(depth, seed, st, td, psy)
        check_golden(T.rd_run_stages_cpu(T.load_hip(depth), T.load_oracle(depth), c), c, gold, k)


@pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref not built")
def test_host_stages_match_reference_fresh_seeds():
    for k, (depth, seed, st, td, psy) in enumerate([(8, 301, 1, 2, 2.0), (8, 302, 0, 3, 0.0), (10, 303, 0, 4, 2.0), (8, 304, 1, 1, 0.4)]):
        c = T.rd_case(depth, seed, st, td, psy, ncu=12)
        want = T.rd_run_ref(T.load_ref(depth), c)
        T.rd_compare(T.rd_run_stages_cpu(T.load_hip(depth), T.load_oracle(depth), c), want, c, "case %d" % k)


def test_outcomes_are_varied():
    """the golden set exercises skips, transform splits, zeroed roots and mixed coded block flags"""
    gold = np.load(GOLD_PATH)
    skip = split = root0 = mixed = 0
    for k, (depth, seed, st, td, psy) in enumerate(CASES):
        c = T.rd_case(depth, seed, st, td, psy)
        for i in range(len(c["cus"])):
            u = gold["%d/%d/units" % (k, i)]
            skip += int(u[0, 4] == 3); split += int(u[:, 0].max() > 0); root0 += int(not u[:, 1:4].any()); mixed += int(u[:, 1:4].any() and not u[:, 1:4].all())
    assert skip > 0 and split > 5 and root0 > 5 and mixed > 5, (skip, split, root0, mixed)


@pytest.mark.gpu
def test_hip_inter_residual_rd_matches_reference_golden():
    gold = np.load(GOLD_PATH)
    for k, (depth, seed, st, td, psy) in enumerate(CASES):
        c = T.rd_case(depth, seed, st, td, psy)
        check_golden(T.rd_run_hip(T.load_hip(depth), c), c, gold, k)


@pytest.mark.gpu
def test_hip_matches_host_stages_on_a_larger_batch():
    """200 candidates in one call (one launch of ~10^4 transform chains) against the staged CPU run of the same host code"""
    c = T.rd_case(8, 401, 0, 3, 2.0, ncu=200)
    got = T.rd_run_hip(T.load_hip(8), c)
    want = T.rd_run_stages_cpu(T.load_hip(8), T.load_oracle(8), c)
    T.rd_compare(got, want, c, "batch")


# ---- Search::encodeResAndCalcRdSkipCU ----
SKIP_CASES = [(8, 501, 1, 2.0), (8, 502, 0, 0.0), (10, 503, 0, 2.0)]
SKIP_GOLD_PATH = os.path.join(T.GOLDEN_DIR, "skip_rd_golden.npz")


@pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref not built")
def test_skip_host_stage_matches_reference():
    for k, (depth, seed, st, psy) in enumerate(SKIP_CASES + [(8, 511, 1, 0.6), (10, 512, 1, 0.0)]):
        c = T.skip_case(depth, seed, st, psy)
        T.rd_compare(T.skip_run_host_cpu(T.load_hip(depth), T.load_oracle(depth), c), T.skip_run_ref(T.load_ref(depth), c), c, "skip case %d" % k)


def test_skip_host_stage_matches_golden():
    gold = np.load(SKIP_GOLD_PATH)
    for k, (depth, seed, st, psy) in enumerate(SKIP_CASES):
        c = T.skip_case(depth, seed, st, psy)
        check_golden(T.skip_run_host_cpu(T.load_hip(depth), T.load_oracle(depth), c), c, gold, k)


@pytest.mark.gpu
def test_hip_skip_rd_matches_reference_golden():
    gold = np.load(SKIP_GOLD_PATH)
    for k, (depth, seed, st, psy) in enumerate(SKIP_CASES):
        c = T.skip_case(depth, seed, st, psy)
        check_golden(T.skip_run_hip(T.load_hip(depth), c), c, gold, k)
