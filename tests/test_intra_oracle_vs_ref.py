"""Pins the intra oracle (oracle/hevc_oracle_intra.c) against the reference's own Predict::initAdiPattern and primitives
driven through oracle/_ref/librefprims*.so.  This container only."""
import numpy as np
import pytest

import hevc_testlib as T

pytestmark = pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref not built (needs /root/reference)")


@pytest.mark.parametrize("depth", [8, 10])
def test_neighbours_and_scan(depth):
    ref, orc = T.load_ref(depth), T.load_oracle(depth)
    strong = 0
    for seed in range(4):
        cases = T.intra_cases(depth, 500 + seed, 200)
        want = T.intra_run_host(ref, cases)
        got = T.intra_run_host(orc, cases)
        for i, (w, g) in enumerate(zip(want, got)):
            info = {k: v for k, v in cases[i].items() if k in ("log2", "strong")}
            assert np.array_equal(w[0], g[0]), (i, info, "ref samples")
            if w[1] is not None:
                assert np.array_equal(w[1], g[1]), (i, info, "filtered samples")
                if cases[i]["log2"] == 5 and cases[i]["strong"]:
                    N2 = 64
                    strong += int(not np.array_equal(w[1][1:N2], ((2 * w[0][1:N2].astype(int) + w[0][0:N2 - 1] + w[0][2:N2 + 1] + 2) >> 2)))
            assert np.array_equal(w[2], g[2]), (i, info, "sa8d scan")
    assert strong > 3       # the bilinear path was taken
