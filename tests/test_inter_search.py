"""Whole-CU inter search (x265amd_pred_inter_search) against golden results of the reference's own Search::predInterSearch run on
fixtures of the same pictures and motion fields (tests/golden/inter_search_golden.npz, generated here from oracle/_ref)."""
import hashlib
import os

import numpy as np
import pytest

import hevc_testlib as T

CASES = [(8, 11, False), (8, 12, True), (8, 13, True), (8, 14, False), (10, 15, True), (10, 16, False), (8, 17, True), (8, 18, True)]
GOLD_PATH = os.path.join(T.GOLDEN_DIR, "inter_search_golden.npz")


def pack(res):
    bits = np.array([r[0] for r in res], np.int32)
    pus = np.concatenate([r[1].view(np.uint8).ravel() for r in res])
    h = hashlib.sha256()
    for r in res:
        for a in r[2:]:
            h.update(np.ascontiguousarray(a).tobytes())
    return bits, pus, np.frombuffer(h.digest(), np.uint8)


@pytest.mark.gpu
def test_hip_pred_inter_search_matches_reference_golden():
    gold = np.load(GOLD_PATH)
    mes = {}
    for i, (depth, seed, b) in enumerate(CASES):
        if depth not in mes:
            mes[depth] = T.HipME(depth)
        c = T.inter_search_case(depth, seed, b)
        bits, pus, dig = pack(T.inter_search_run_hip(mes[depth].L if hasattr(mes[depth], "L") else T.load_hip(depth), mes[depth], c))
        assert np.array_equal(bits, gold["bits/%d" % i]), (i, bits, gold["bits/%d" % i])
        assert np.array_equal(pus, gold["pus/%d" % i]), (i, np.nonzero(pus != gold["pus/%d" % i])[0][:8] // 24)
        assert np.array_equal(dig, gold["pred/%d" % i]), i
