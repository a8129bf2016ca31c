"""Pin the oracle against golden vectors generated from the reference build (tests/golden/make_golden.py).
Runs anywhere (no /root/reference, no GPU)."""
import json
import os

import numpy as np
import pytest

import hevc_testlib as T

with open(os.path.join(T.GOLDEN_DIR, "prims_digests.json")) as f:
    DIGESTS = json.load(f)["digests"]
REPS = {"random": 3, "min": 1, "max": 1}


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("name", sorted(T.CASES))
def test_digests(name, depth):
    orc = T.load_oracle(depth)
    for mode in T.MODES:
        for rep in range(REPS[mode]):
            key = "%s/%d/%s/%d" % (name, depth, mode, rep)
            assert T.digest(T.run_case(orc, name, mode, rep)) == DIGESTS[key], key


@pytest.mark.parametrize("depth", [8, 10])
def test_full_vectors(depth):
    orc = T.load_oracle(depth)
    gold = np.load(os.path.join(T.GOLDEN_DIR, "prims_%d.npz" % depth))
    for name in sorted(T.CASES):
        outs = T.run_case(orc, name, "random", 0)
        want = [gold["%s/%04d" % (name, i)] for i in range(len(outs))]
        T.assert_same(outs, want, "%s/%d" % (name, depth))
