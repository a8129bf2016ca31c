"""Oracle (oracle/hevc_oracle_entropy.c) against the reference itself (oracle/_ref/librefprims*.so): CABAC tables, context
initialisation, estBit tables, RDOQ (Quant::rdoQuant through transformNxN) and bits-only coefficient coding."""
import ctypes as C

import numpy as np
import pytest

import hevc_testlib as T

pytestmark = pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref not built (needs /root/reference)")


@pytest.mark.parametrize("depth", [8, 10])
def test_cabac_tables(depth):
    R, O = T.load_ref(depth), T.load_oracle(depth)
    bits = np.zeros(128, np.uint32); nxt = np.zeros((128, 2), np.uint8)
    R.lib.ref_ctx_tables(T._ptr(bits), T._ptr(nxt))
    O.lib.orc_ctx_bits.restype = C.c_uint32
    O.lib.orc_ctx_next.restype = C.c_uint8
    for s in range(128):
        for b in range(2):
            assert O.lib.orc_ctx_next(C.c_uint8(s), b) == nxt[s, b], (s, b)
            assert O.lib.orc_ctx_bits(C.c_uint8(s), b) == bits[s ^ b]


@pytest.mark.parametrize("depth", [8, 10])
def test_context_reset_and_est_bit(depth):
    R, O = T.load_ref(depth), T.load_oracle(depth)
    assert R.lib.ref_est_bits_ints() == T.EST_INTS
    for st in range(3):
        for qp in range(0, 52):
            a, b = T.entropy_reset(R, st, qp), T.entropy_reset(O, st, qp)
            assert np.array_equal(a, b), (st, qp)
            if qp % 5 == 0:
                for log2 in range(2, 6):
                    for luma in (1, 0):
                        if not luma and log2 == 5:
                            continue
                        assert np.array_equal(T.est_bit(R, a, log2, luma), T.est_bit(O, a, log2, luma)), (st, qp, log2, luma)


@pytest.mark.parametrize("depth", [8, 10])
def test_est_bit_adapted_contexts(depth):
    """context states away from their initial values (as after coding some CTUs)"""
    R, O = T.load_ref(depth), T.load_oracle(depth)
    rng = np.random.default_rng(5)
    for i in range(40):
        ctx = rng.integers(0, 126, T.CTX_COUNT).astype(np.uint8)
        log2, luma = int(rng.integers(2, 6)), int(rng.integers(0, 2))
        assert np.array_equal(T.est_bit(R, ctx, log2, luma), T.est_bit(O, ctx, log2, luma))


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_rdoq(depth, seed):
    R, O = T.load_ref(depth), T.load_oracle(depth)
    cases = T.rdoq_cases(depth, seed, 250)
    a, b = T.rdoq_run(R, cases), T.rdoq_run(O, cases)
    nz = 0
    for i, (x, y) in enumerate(zip(a, b)):
        assert x[0] == y[0] and np.array_equal(x[1], y[1]), (i, {k: v for k, v in cases[i].items() if k not in ("fenc", "pred", "ctx")}, x[0], y[0])
        nz += x[0] > 0
    assert nz > len(cases) // 3


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("seed", [4, 5])
def test_coeff_bits(depth, seed):
    R, O = T.load_ref(depth), T.load_oracle(depth)
    cases = T.rdoq_cases(depth, seed, 300)
    levels = T.rdoq_run(O, cases)
    a, b = T.coeff_bits_run(R, cases, levels), T.coeff_bits_run(O, cases, levels)
    n = 0
    for i, (x, y) in enumerate(zip(a, b)):
        assert x[0] == y[0] and np.array_equal(x[1], y[1]), (i, x[0], y[0])
        n += x[0] > 0
    assert n > len(cases) // 3
