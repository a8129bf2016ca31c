"""GPU parity: every per-slot C-ABI entry point of libx265amd (HIP kernels, gfx950) against the CPU oracle on the
same seeded inputs, and against the golden digests generated from the reference build.  TestBench pattern
(reference: source/test/testbench.cpp:102-261): random / min / max buffers, exact equality -- integer paths, so the
tolerance is zero."""
import json
import os

import pytest

import hevc_testlib as T

pytestmark = pytest.mark.gpu

with open(os.path.join(T.GOLDEN_DIR, "prims_digests.json")) as f:
    DIGESTS = json.load(f)["digests"]
REPS = {"random": 3, "min": 1, "max": 1}


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("name", sorted(T.CASES))
def test_slot_parity(name, depth):
    hip, orc = T.load_hip(depth), T.load_oracle(depth)
    assert hip.lib.x265amd_device_count() >= 1, "no GPU visible: the HIP path must not silently fall back"
    for mode in T.MODES:
        for rep in range(REPS[mode]):
            want = T.run_case(orc, name, mode, rep)
            got = T.run_case(hip, name, mode, rep)
            key = "%s/%d/%s/%d" % (name, depth, mode, rep)
            T.assert_same(got, want, key)
            assert T.digest(got) == DIGESTS[key], "golden digest mismatch " + key
