"""Merge / AMVP candidate derivation (host code of the product) against the reference's own CUData methods on fixtures built from
the same raster motion fields (oracle/_ref), and against a committed digest."""
import hashlib
import json
import os

import numpy as np
import pytest

import hevc_testlib as T

CASES = [(1, 192, 128, True), (2, 256, 192, False), (3, 200, 136, True), (4, 128, 128, True), (5, 320, 64, False), (6, 136, 200, True)]


def digest(results):
    h = hashlib.sha256()
    for merge, res in results:
        h.update(merge.tobytes())
        for key in sorted(res):
            h.update(res[key][0].tobytes()); h.update(res[key][1].tobytes())
    return h.hexdigest()


@pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_product_vs_reference():
    R, P = T.load_ref(8), T.load_hip(8)
    nmerge = nmvc = 0
    for (seed, w, h, b) in CASES:
        for rep in range(3):
            c = T.mvpred_case(seed + 50 * rep, w, h, b)
            a, p = T.mvpred_run(R, c), T.mvpred_run(P, c)
            for i, ((ma, ra), (mp, rp)) in enumerate(zip(a, p)):
                assert len(ma) == len(mp) and ma.tobytes() == mp.tobytes(), (seed, rep, c["pus"][i], ma, mp)
                for key in ra:
                    assert np.array_equal(ra[key][0], rp[key][0]), (seed, rep, c["pus"][i], key, "amvp", ra[key][0], rp[key][0])
                    assert np.array_equal(ra[key][1], rp[key][1]), (seed, rep, c["pus"][i], key, "mvc", ra[key][1], rp[key][1])
                    nmvc += len(ra[key][1])
                nmerge += len(ma)
    assert nmerge > 1000 and nmvc > 3000


def test_product_matches_golden():
    with open(os.path.join(T.GOLDEN_DIR, "mvpred_golden.json")) as f:
        gold = json.load(f)
    P = T.load_hip(8)
    for i, (seed, w, h, b) in enumerate(CASES):
        assert digest(T.mvpred_run(P, T.mvpred_case(seed, w, h, b))) == gold[str(i)]
