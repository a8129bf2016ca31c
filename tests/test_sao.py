"""Sample adaptive offset over whole pictures -- per-CTU statistics and the application of given parameters: the oracle against
the reference's own SAO class (oracle/_ref), a committed digest, and the GPU kernels against the oracle."""
import hashlib
import json
import os

import numpy as np
import pytest

import hevc_testlib as T

CASES = [(1, 200, 136), (2, 128, 128), (3, 264, 72), (4, 64, 200), (5, 136, 192)]


def digest(results):
    h = hashlib.sha256()
    for cnt, org, out in results:
        h.update(cnt.tobytes()); h.update(org.tobytes())
        for p in out:
            h.update(p.tobytes())
    return h.hexdigest()


@pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("depth", [8, 10])
def test_oracle_vs_reference(depth):
    R, O = T.load_ref(depth), T.load_oracle(depth)
    for (seed, w, h) in CASES:
        c = T.sao_case(depth, seed, w, h)
        a, b = T.sao_run_host(R, c), T.sao_run_host(O, c)
        assert np.array_equal(a[0], b[0]), (seed, "count", np.nonzero(a[0] != b[0])[0][:8])
        assert np.array_equal(a[1], b[1]), (seed, "offsetOrg")
        for k in range(3):
            assert np.array_equal(a[2][k], b[2][k]), (seed, "apply", k, np.argwhere(a[2][k] != b[2][k])[:5])
        assert a[0].sum() > 1000 and sum(int((x != y).sum()) for x, y in zip(a[2], c["rec"])) > 1000


@pytest.mark.parametrize("depth", [8, 10])
def test_oracle_matches_golden(depth):
    O = T.load_oracle(depth)
    with open(os.path.join(T.GOLDEN_DIR, "sao_golden.json")) as f:
        gold = json.load(f)
    assert digest([T.sao_run_host(O, T.sao_case(depth, *c)) for c in CASES]) == gold[str(depth)]


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [8, 10])
def test_hip_sao(depth):
    H, O = T.load_hip(depth), T.load_oracle(depth)
    with open(os.path.join(T.GOLDEN_DIR, "sao_golden.json")) as f:
        gold = json.load(f)
    res = []
    for c in CASES + [(7, 1920, 1080 // 8 * 8)]:
        case = T.sao_case(depth, *c)
        got, want = T.sao_run_hip(H, case), T.sao_run_host(O, case)
        assert np.array_equal(got[0], want[0]), (c, "count")
        assert np.array_equal(got[1], want[1]), (c, "offsetOrg")
        for k in range(3):
            assert np.array_equal(got[2][k], want[2][k]), (c, "apply", k)
        res.append(got)
    assert digest(res[:len(CASES)]) == gold[str(depth)]
