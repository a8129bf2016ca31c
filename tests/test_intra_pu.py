"""x265amd_intra_pu (include/x265amd.h): one prediction unit of Search::estIntraPredQT -- 35-mode scan, candidate list, the candidates' transform chains -- as
one launch.  Against the oracle: the scan (hevc_oracle_intra.c, pinned against the reference's Predict class), the candidate list restated from
search.cpp:1615-1650 / :3953-3972 on the oracle's costs, and the oracle's intra TU step for every listed mode."""
import numpy as np
import pytest

import hevc_testlib as T


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [8, 10])
def test_hip_intra_pu(depth):
    H, O = T.load_hip(depth), T.load_oracle(depth)
    rng = np.random.default_rng(99 + depth)
    cases = [c for c in T.intra_tu_cases(depth, 3100, 160) if c["log2"] <= 5]
    seen_full = seen_short = 0
    for c in cases:
        c["ttype"], c["rdoq"] = 0, 0
        preds = [int(x) for x in rng.choice(35, 3, replace=False)]
        rbits, mpm_base = int(rng.integers(5, 9)), int(rng.integers(0, 3))
        lam = int(rng.integers(200, 60000))
        max_cand = int(rng.integers(3, 11))
        want_sa8d = T.intra_run_host(O, [c])[0][2]
        sa8d, modes, per = T.intra_pu_run_hip(H, c, preds, rbits, mpm_base, lam, max_cand)
        assert np.array_equal(sa8d, want_sa8d), ("sa8d", c["log2"])
        want_modes = T.intra_pu_candidates(want_sa8d, preds, rbits, mpm_base, lam, max_cand)
        assert modes == want_modes, (c["log2"], modes, want_modes)
        seen_full += len(modes) == max_cand; seen_short += len(modes) < max_cand
        per_mode = []
        for m in modes:
            cm = dict(c); cm["mode"] = m
            per_mode.append(cm)
        want = T.intra_tu_run_host(O, per_mode)
        for i, (g, w) in enumerate(zip(per, want)):
            assert g[0] == w[0], (c["log2"], modes[i], g[0], w[0])
            for k in range(1, 5):
                assert np.array_equal(np.asarray(g[k]).reshape(-1), np.asarray(w[k]).reshape(-1)), (c["log2"], modes[i], k)
    assert seen_full > 10 and seen_short > 10, "both the full-list and the short-list path must be exercised"
