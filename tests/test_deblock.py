"""In-loop deblocking of whole pictures: the oracle against the reference's own Deblock class on CUData fixtures built from
random coding quad-trees (oracle/_ref), a committed digest, and the GPU kernels against the oracle."""
import hashlib
import json
import os

import numpy as np
import pytest

import hevc_testlib as T

CASES = [(1, 200, 136, True, False), (2, 128, 128, False, False), (3, 264, 72, True, True), (4, 64, 192, True, False), (5, 136, 200, False, True)]


def digest(planes_list):
    h = hashlib.sha256()
    for pl in planes_list:
        for p in pl:
            h.update(p.tobytes())
    return h.hexdigest()


@pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("depth", [8, 10])
def test_oracle_vs_reference(depth):
    R, O = T.load_ref(depth), T.load_oracle(depth)
    changed = 0
    for (seed, w, h, b, byp) in CASES:
        c = T.deblock_case(depth, seed, w, h, b, byp)
        for passes in (1, 3):
            a, o = T.deblock_run_host(R, c, passes), T.deblock_run_host(O, c, passes)
            for k in range(3):
                assert np.array_equal(a[k], o[k]), (seed, passes, k, np.argwhere(a[k] != o[k])[:5])
        changed += int((a[0] != c["planes"][0]).sum()) + int((a[1] != c["planes"][1]).sum())
    assert changed > 5000


@pytest.mark.parametrize("depth", [8, 10])
def test_oracle_matches_golden(depth):
    O = T.load_oracle(depth)
    with open(os.path.join(T.GOLDEN_DIR, "deblock_golden.json")) as f:
        gold = json.load(f)
    assert digest([T.deblock_run_host(O, T.deblock_case(depth, *c)) for c in CASES]) == gold[str(depth)]


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [8, 10])
def test_hip_deblock(depth):
    H, O = T.load_hip(depth), T.load_oracle(depth)
    with open(os.path.join(T.GOLDEN_DIR, "deblock_golden.json")) as f:
        gold = json.load(f)
    outs = []
    for c in CASES:
        case = T.deblock_case(depth, *c)
        got, want = T.deblock_run_hip(H, case), T.deblock_run_host(O, case)
        for k in range(3):
            assert np.array_equal(got[k], want[k]), (c, k, np.argwhere(got[k] != want[k])[:5])
        outs.append(got)
    assert digest(outs) == gold[str(depth)]
    big = T.deblock_case(depth, 9, 1920, 1080 // 8 * 8, True, False)          # a full-size picture
    got, want = T.deblock_run_hip(H, big), T.deblock_run_host(O, big)
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
