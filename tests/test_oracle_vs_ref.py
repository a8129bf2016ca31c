"""Pin the oracle (oracle/hevc_oracle.c) against the REFERENCE ITSELF (oracle/_ref/librefprims*.so, built from
/root/reference by oracle/build_ref.sh).  TestBench pattern (reference: source/test/testbench.cpp:102-261):
random / min / max buffers, exact equality.  Skipped where oracle/_ref is absent; test_oracle_golden.py then
still pins the oracle against vectors generated from the same reference build."""
import numpy as np
import pytest

import hevc_testlib as T

pytestmark = pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref not built (needs /root/reference)")


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("name", sorted(T.CASES))
def test_case(name, depth):
    ref, orc = T.load_ref(depth), T.load_oracle(depth)
    for mode in T.MODES:
        for rep in range(3 if mode == "random" else 1):
            want = T.run_case(ref, name, mode, rep)
            got = T.run_case(orc, name, mode, rep)
            T.assert_same(got, want, "%s/%d/%s/%d" % (name, depth, mode, rep))


@pytest.mark.parametrize("depth", [8, 10])
def test_tables(depth):
    import ctypes as C
    ref, orc = T.load_ref(depth), T.load_oracle(depth)

    def arr(lib, fn, n, ty, *args):
        f = getattr(lib.lib, fn)
        f.restype = C.POINTER(ty)
        p = f(*args)
        return np.array([p[i] for i in range(n)])

    for log2n, name in ((2, "t4"), (3, "t8"), (4, "t16"), (5, "t32")):
        n = 1 << log2n
        assert np.array_equal(arr(ref, "ref_tbl_" + name, n * n, C.c_int16), arr(orc, "orc_tbl_dct", n * n, C.c_int16, log2n))
    assert np.array_equal(arr(ref, "ref_tbl_lumaFilter", 32, C.c_int16), arr(orc, "orc_tbl_lumaFilter", 32, C.c_int16))
    assert np.array_equal(arr(ref, "ref_tbl_chromaFilter", 32, C.c_int16), arr(orc, "orc_tbl_chromaFilter", 32, C.c_int16))
    flags = arr(ref, "ref_tbl_intraFilterFlags", 35, C.c_uint8)
    assert [orc.lib.orc_intra_filter_flags(m) for m in range(35)] == list(flags)
    for w, h in T.PU_SIZES:
        assert ref.lib.ref_partition_from_sizes(w, h) == orc.lib.orc_partition_from_sizes(w, h) == T.PU_SIZES.index((w, h))
