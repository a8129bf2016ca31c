"""Intra TU coding step (codeIntraLumaQT / codeIntraChromaQt per TU): one-mode neighbour set + prediction + residual chain.
Oracle against the reference's own Predict / Quant classes (oracle/_ref) and committed golden digests; the fused GPU kernel
(x265amd_intra_tu_chain) against the oracle."""
import hashlib
import os

import numpy as np
import pytest

import hevc_testlib as T

GOLD_PATH = os.path.join(T.GOLDEN_DIR, "intra_tu_golden.npz")


def digest(res):
    h = hashlib.sha256()
    for st, pred, recon, coeff, resi in res:
        h.update(np.array(st, np.uint64).tobytes())
        for a in (pred, recon, coeff, resi):
            h.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(h.digest(), np.uint8)


def same(a, b):
    for i, (x, y) in enumerate(zip(a, b)):
        assert x[0] == y[0], (i, x[0], y[0])
        for k in range(1, 5):
            assert np.array_equal(x[k], y[k]), (i, k)


@pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("depth", [8, 10])
def test_oracle_vs_reference(depth):
    R, O = T.load_ref(depth), T.load_oracle(depth)
    for seed in range(2):
        cases = T.intra_tu_cases(depth, 2100 + seed, 300, rdoq=bool(seed))
        pa, pb = T.intra_predict_host(R, cases), T.intra_predict_host(O, cases)
        for i, (x, y) in enumerate(zip(pa, pb)):
            assert np.array_equal(x, y), (i, cases[i]["log2"], cases[i]["ttype"], cases[i]["mode"])
        same(T.intra_tu_run_host(R, cases), T.intra_tu_run_host(O, cases))


@pytest.mark.parametrize("depth", [8, 10])
def test_oracle_matches_golden(depth):
    gold = np.load(GOLD_PATH)
    O = T.load_oracle(depth)
    for seed in range(2):
        cases = T.intra_tu_cases(depth, 2100 + seed, 300, rdoq=bool(seed))
        assert np.array_equal(digest(T.intra_tu_run_host(O, cases)), gold["digest/%d/%d" % (depth, seed)])


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [8, 10])
def test_hip_intra_tu_chain(depth):
    gold = np.load(GOLD_PATH)
    H, O = T.load_hip(depth), T.load_oracle(depth)
    for seed in range(3):
        cases = T.intra_tu_cases(depth, 2100 + seed, 300, rdoq=bool(seed))
        got = T.intra_tu_run_hip(H, cases)
        same(got, T.intra_tu_run_host(O, cases))
        if seed < 2:
            assert np.array_equal(digest(got), gold["digest/%d/%d" % (depth, seed)])
