"""Final entropy coding of CTUs (the CABAC write pass): the product's host coder (x265amd_cabac_*) against the reference's own
Entropy::encodeCTU ... finishSlice on CUData built from the same decisions (oracle/_ref), and against committed golden bitstreams."""
import hashlib
import json
import os

import numpy as np
import pytest

import hevc_testlib as T

CASES = [(1, 128, 128, 2), (2, 200, 136, 1), (3, 264, 72, 0), (4, 64, 64, 0), (5, 136, 200, 1), (6, 192, 128, 0), (7, 128, 64, 2), (8, 320, 192, 0)]


@pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("depth", [8, 10])
def test_product_vs_reference(depth):
    R, P = T.load_ref(depth), T.load_hip(depth)
    total = 0
    for (seed, w, h, st) in CASES:
        for dense in (False, True):
            c = T.cabac_case(seed + 100 * dense, w, h, st, dense)
            a, b = T.cabac_run_ref(R, c), T.cabac_run_product(P, c)
            assert len(a[0]) == len(b[0]) and np.array_equal(a[0], b[0]), (seed, dense, len(a[0]), len(b[0]), int(np.argmax(a[0][:min(len(a[0]), len(b[0]))] != b[0][:min(len(a[0]), len(b[0]))])))
            assert np.array_equal(a[1], b[1]), (seed, "contexts")
            assert np.array_equal(a[2], b[2]), (seed, "qp map")
            total += len(a[0])
            # bit-counting mode: same context evolution
            a2, b2 = T.cabac_run_ref(R, c, 1), T.cabac_run_product(P, c, 1)
            assert np.array_equal(a2[1], b2[1]) and np.array_equal(a2[1], a[1])
    assert total > 20000


def case_digest(res):
    h = hashlib.sha256()
    h.update(res[0].tobytes()); h.update(res[1].tobytes()); h.update(res[2].tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("depth", [8, 10])
def test_product_matches_golden(depth):
    """host code only: runs wherever the library loads, the golden digests come from the reference's Entropy class"""
    with open(os.path.join(T.GOLDEN_DIR, "cabac_golden.json")) as f:
        gold = json.load(f)[str(depth)]
    P = T.load_hip(depth)
    for i, (seed, w, h, st) in enumerate(CASES):
        for dense in (False, True):
            c = T.cabac_case(seed + 100 * dense, w, h, st, dense)
            assert case_digest(T.cabac_run_product(P, c)) == gold["%d/%d" % (i, int(dense))]


@pytest.mark.parametrize("depth", [8])
def test_bit_counting_mode_tracks_the_bitstream(depth):
    """fractional bits accumulated per CTU in bit-counting mode against the real slice size: CABAC estimates are within a few percent"""
    import ctypes as C
    P = T.load_hip(depth)
    c = T.cabac_case(8, 320, 192, 0, True)
    real = len(T.cabac_run_product(P, c)[0]) * 8
    units = c["units"].copy()
    si = np.array([c["si"]], T.SLICE_INFO_DT)
    P.lib.x265amd_cabac_open.restype = C.c_void_p
    P.lib.x265amd_cabac_ctu_bits.restype = C.c_uint64
    h = C.c_void_p(P.lib.x265amd_cabac_open(T._ptr(si), T._ptr(units), 1))
    est = 0
    for a in range(c["ctus"]):
        row = c["coeff"][a]
        assert P.lib.x265amd_cabac_encode_ctu(h, a, T.off(row, 0), T.off(row, 64 * 64), T.off(row, 64 * 64 + 32 * 32)) == 0
        est += P.lib.x265amd_cabac_ctu_bits(h)
    P.lib.x265amd_cabac_close(h)
    assert abs(est / 32768.0 - real) < 0.03 * real, (est / 32768.0, real)
