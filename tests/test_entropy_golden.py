"""Oracle against golden results of the reference's Entropy / Quant classes (tests/golden/entropy_golden.npz, written by
tests/golden/make_golden.py from oracle/_ref): context initialisation, estBit tables, RDOQ levels, bits-only coefficient coding."""
import os

import numpy as np
import pytest

import hevc_testlib as T

GOLD = np.load(os.path.join(T.GOLDEN_DIR, "entropy_golden.npz"))


def check_entropy(L, depth, seeds=(0, 1, 2)):
    assert np.array_equal(np.stack([T.entropy_reset(L, st, qp) for st in range(3) for qp in range(52)]), GOLD["reset/%d" % depth])
    ests = []
    for st in range(3):
        for qp in (0, 17, 30, 43, 51):
            ctx = GOLD["reset/%d" % depth][st * 52 + qp]
            for log2 in range(2, 6):
                for luma in (1, 0):
                    if luma or log2 < 5:
                        ests.append(T.est_bit(L, ctx, log2, luma))
    assert np.array_equal(np.stack(ests), GOLD["est/%d" % depth])
    for seed in seeds:
        cases = T.rdoq_cases(depth, 700 + seed, 250)
        res = T.rdoq_run(L, cases)
        assert np.array_equal(np.array([r[0] for r in res], np.int32), GOLD["rdoq/%d/%d/numsig" % (depth, seed)])
        assert np.array_equal(np.concatenate([r[1] for r in res]), GOLD["rdoq/%d/%d/coeff" % (depth, seed)])
        bits = T.coeff_bits_run(L, cases, res)
        assert np.array_equal(np.array([b[0] for b in bits], np.uint64), GOLD["rdoq/%d/%d/bits" % (depth, seed)])
        assert np.array_equal(np.stack([b[1] for b in bits]), GOLD["rdoq/%d/%d/ctx" % (depth, seed)])


@pytest.mark.parametrize("depth", [8, 10])
def test_oracle_entropy_matches_golden(depth):
    check_entropy(T.load_oracle(depth), depth)
