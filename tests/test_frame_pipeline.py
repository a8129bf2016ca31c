"""Frame pipeline end to end against the reference ENCODER (not a fixture): the reference's own command line program encodes a 4-frame
synthetic clip (I P P P; CQP, no AQ / cutree / weighted prediction / in-loop filters / WPP, preset medium otherwise: rd 3, hex subme 2,
3 references, limit-refs 3, early skip, rskip, psy-rd 2.0, sign hiding, strong intra smoothing, temporal MVP); its reconstructed frames
(--recon) and the slice data of its bitstream are the golden data (tests/golden/frame_pipeline_golden.npz, generated here by
tests/golden/make_golden.py).  x265amd_analyse_frame must reproduce every reconstructed sample and every slice-data byte."""
import os

import numpy as np
import pytest

import hevc_testlib as T

GOLD_PATH = os.path.join(T.GOLDEN_DIR, "frame_pipeline_golden.npz")


def test_golden_is_a_real_encode():
    g = np.load(GOLD_PATH)
    assert list(g["slice_qp"]) == [27, 30, 30, 30] and list(g["bframes/slice_qp"]) == [27, 30, 32, 32, 30, 32, 32]
    assert all(len(g["slice/%d" % k]) > 20 for k in range(4))


TAGS = ["", "deblock/", "wpp/", "bframes/", "sao/", "sao_bframes/", "rectamp_bframes/", "rectamp_lm/", "rd5_bframes/", "rd6_rectamp/", "rd2_bframes/", "rd2_rectamp/"]


def header_config(tag):
    return dict(bframes=2 if "bframes" in tag else 0, deblock=bool(tag), wpp=tag == "wpp/", sao="sao" in tag, amp="rectamp" in tag)


@pytest.mark.parametrize("tag", TAGS)
def test_stream_headers_match_reference_encoder(tag):
    """VPS / SPS / PPS NAL units (x265amd_write_stream_headers, host code) against the head of the reference encoder's stream"""
    g = np.load(GOLD_PATH)
    stream = g[tag + "stream"]
    n = len(g[tag + "schedule"])
    first = len(stream) - sum(len(g[tag + "nal/%d" % k]) + 4 for k in range(n))      # every slice NAL opens an access unit: 4-byte start codes
    got = T.frame_stream_headers(T.load_hip(8), **header_config(tag))
    assert got.tobytes().hex() == stream[:first].tobytes().hex()


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_hip_frame_pipeline_matches_reference_encoder(tag):
    """tag "deblock/": the same encode with the in-loop deblocking filter on (x265amd_deblock_units + x265amd_deblock_picture per frame);
    "wpp/": deblocking and wavefront parallel processing on (per-row entropy states, one sub-stream per CTU row, entry points in the slice header);
    "bframes/": a 7-frame clip coded I P b b P b b (two lists, bi-prediction, collocated picture from list 1, non-referenced pictures), deblocking.
    "sao/", "sao_bframes/": the same clips with sample adaptive offset on (x265amd_sao_stats, x265amd_sao_rdo, x265amd_sao_apply, SAO syntax).
    "rectamp_*": --rect --amp (and --limit-modes) on both clips.  "rd5_*", "rd6_*": --rd 5 / 6 (compressInterCU_rd5_6); "rd2_*": --rd 2.
    The slice NAL units (x265amd_write_slice_nal) behind the reference's parameter sets must give the reference's byte stream."""
    import hashlib
    g = np.load(GOLD_PATH)
    me = T.HipME(8)
    sched = g[tag + "schedule"]
    n = len(sched)
    got = T.frame_pipeline_run_hip(T.load_hip(8), me, [int(q) for q in g[tag + "slice_qp"]], nframes=n, deblock=bool(tag), wpp=tag == "wpp/", schedule=sched,
                                   frames=T.frame_clip_b(8) if "bframes" in tag else None, sao="sao" in tag,
                                   rect=int("rectamp" in tag), amp=int("rectamp" in tag), limit_modes=int("_lm" in tag or "rd6" in tag or "rd2_rect" in tag),
                                   rd_level=5 if "rd5" in tag else 6 if "rd6" in tag else 2 if "rd2" in tag else 3)
    for k, (poc, planes, data) in enumerate(got):
        for p in range(3):
            want = g[tag + "recon/%d/%d" % (poc, p)]
            if not np.array_equal(planes[p], want):
                bad = np.argwhere(planes[p] != want)
                raise AssertionError("coded frame %d (poc %d) plane %d: %d reconstructed samples differ from the reference encoder's, first at (y, x) = %s" % (
                    k, poc, p, len(bad), bad[0].tolist()))
        want_nal = g[tag + "nal/%d" % k]
        sc = 4
        assert np.array_equal(data[sc:], want_nal), "coded frame %d: slice NAL unit differs from the reference encoder's (%d vs %d bytes)" % (k, len(data) - sc, len(want_nal))
    # the whole byte stream: our VPS / SPS / PPS (x265amd_write_stream_headers) + our slice NAL units
    stream = g[tag + "stream"]
    ours = np.concatenate([T.frame_stream_headers(T.load_hip(8), **header_config(tag))] + [d for (_, _, d) in got])
    assert len(ours) == len(stream)
    assert hashlib.md5(ours.tobytes()).hexdigest() == hashlib.md5(stream.tobytes()).hexdigest()
