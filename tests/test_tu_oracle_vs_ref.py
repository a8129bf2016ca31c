"""Pins the residual-path oracle (oracle/hevc_oracle_tu.c) against the reference's own Quant and RDCost classes and
scan tables, driven through oracle/_ref/librefprims*.so.  This container only."""
import ctypes as C

import numpy as np
import pytest

import hevc_testlib as T

pytestmark = pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref not built (needs /root/reference)")


@pytest.mark.parametrize("depth", [8, 10])
def test_scan_tables(depth):
    ref, orc = T.load_ref(depth), T.load_oracle(depth)
    ref.lib.ref_tbl_scan.restype = C.POINTER(C.c_uint16)
    orc.lib.orc_tbl_scan.restype = C.POINTER(C.c_uint16)
    for t in range(3):
        for l in range(2, 6):
            n = 1 << (2 * l)
            a = np.ctypeslib.as_array(ref.lib.ref_tbl_scan(t, l), (n,))
            b = np.ctypeslib.as_array(orc.lib.orc_tbl_scan(t, l), (n,))
            assert np.array_equal(a, b), (t, l)


@pytest.mark.parametrize("depth", [8, 10])
def test_lambda2_table(depth):
    ref, orc = T.load_ref(depth), T.load_oracle(depth)
    ref.lib.ref_tbl_lambda2.restype = C.POINTER(C.c_double)
    orc.lib.orc_lambda2.restype = C.c_double
    t = ref.lib.ref_tbl_lambda2()
    for qp in range(70):
        assert orc.lib.orc_lambda2(qp) == t[qp], qp


@pytest.mark.parametrize("depth", [8, 10])
def test_transform_inverse(depth):
    ref, orc = T.load_ref(depth), T.load_oracle(depth)
    for seed in range(4):
        cases = T.tu_cases(depth, 100 + seed, 150)
        want = T.tu_run_host(ref, cases)
        got = T.tu_run_host(orc, cases)
        nz = 0
        for i, (w, g) in enumerate(zip(want, got)):
            assert w[0] == g[0] and np.array_equal(w[1], g[1]) and np.array_equal(w[2], g[2]), (i, {k: v for k, v in cases[i].items() if k not in ("fenc", "pred")}, w[0], g[0])
            nz += w[0] > 1
        assert nz > 40      # sign hiding actually exercised


@pytest.mark.parametrize("depth", [8, 10])
def test_rdcost(depth):
    ref, orc = T.load_ref(depth), T.load_oracle(depth)
    rng = np.random.default_rng(5)
    for _ in range(300):
        qp, st = int(rng.integers(0, 70)), int(rng.integers(0, 3))
        psy = float(rng.choice([0.0, 1.0, 2.0, 0.7]))
        dist, bits, pc = int(rng.integers(0, 1 << 24)), int(rng.integers(0, 1 << 16)), int(rng.integers(0, 1 << 16))
        a = np.zeros(6, np.uint64); b = np.zeros(6, np.uint64)
        ref.lib.ref_rdcost(qp, st, C.c_double(psy), C.c_uint64(dist), C.c_uint32(bits), C.c_uint32(pc), T._ptr(a))
        orc.lib.orc_rdcost(qp, st, C.c_double(psy), C.c_uint64(dist), C.c_uint32(bits), C.c_uint32(pc), T._ptr(b))
        assert np.array_equal(a, b), (qp, st, psy, a, b)
