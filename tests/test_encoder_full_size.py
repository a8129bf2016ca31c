"""BASELINE.json's configurations 3, 4 and 5 at their STATED size through the encoder object (include/x265amd_encoder.h), against the reference encoder's stream
and reconstruction digests (tests/golden/encoder_full_golden.json, made by tests/golden/make_golden.py full with oracle/_ref/x265_ref{8,10}):

  cfg3  3840x2160  8-bit  x265 --preset slow --qp 30 --no-info    (rd 4, RDOQ 2, star / subme 3, 4 references, rect + limit-modes), 12 frames
  cfg4  3840x2160 10-bit  x265 --preset medium --qp 30 --no-info  (Main 10),                                                      12 frames
  fhd_medium_60  1920x1080  x265 --preset medium --qp 30 --no-info, 60 frames: the bench's configuration over both re-seeds of the clip (scene cuts at 24 and 48)
  fhd_rd2  1920x1080 --preset medium --rd 2 --bframes 1: Analysis::complexityCheckCU (analysis.cpp:3536-3559) is active only for rd 0-2 on pictures of >= 1080 rows
  cfg5  7680x4320 10-bit  x265 --preset veryslow --rd 6 --qp 30 --no-info, 3 frames: the preset as it comes since round 5 (--weightb is coded; AMP, TU depth 3,
        5 merge candidates, 5 references, subme 4, eight B frames / forty pictures of lookahead in the decision)

All run the reference's presets as they come: the trellis of --b-adapt 2, scene-cut detection, open GOPs, B pyramid, lookahead slices, weighted prediction
(no weight is chosen on these clips), frame-parallel rules (the reference picks three frame threads on the eight cores the fixtures were made on).
Picture-size dependent code is what these pin: 34 / 68 CTU rows in flight and the queues they take, the cut last CTU row (2160 = 33 x 64 + 48, 4320 = 67 x 64 + 32),
64-bit squared errors of Main 10 (common.h:142-146), the star search's raster over the full window, level / DPB derivation of the headers at these sizes; since
round 5 cfg3 and cfg4 cover whole mini-GOPs of the trellis with the B pyramid at 2160p (twelve pictures each).
The clip is SURVEY.md section 8d's generator (hevc_testlib.survey_clip)."""
import hashlib
import json
import os

import numpy as np
import pytest

import hevc_testlib as T

GOLD = os.path.join(T.GOLDEN_DIR, "encoder_full_golden.json")
PRESET_GOLD = os.path.join(T.GOLDEN_DIR, "encoder_preset_golden.json")
RC_GOLD = os.path.join(T.GOLDEN_DIR, "encoder_rc_golden.json")


def test_full_size_golden_present():
    g = json.load(open(GOLD))
    for tag, ((w, h), n, depth, cfg_id, cfg, cli) in T.FULL_CASES.items():
        assert tag in g and len(g[tag]["recon_md5"]) == n and g[tag]["stream_bytes"] > 0, tag


def test_survey_clip_is_the_bench_clip():
    """the generator is integer only and depends on (t, cfg_id) alone; 10-bit samples are the 8-bit ones times four"""
    a = T.survey_clip(256, 128, 8, 2, 23, 2)
    b = T.survey_clip(256, 128, 10, 2, 23, 2)
    for fa, fb in zip(a, b):
        for pa, pb in zip(fa, fb):
            assert pa.dtype == np.uint8 and pb.dtype == np.uint16 and np.array_equal(pa.astype(np.uint16) * 4, pb)
    assert not np.array_equal(a[0][0], a[1][0])         # frame 24 is a new scene
    # The noise field is SURVEY.md section 8d's 32-bit LCG (hevc_testlib.lcg_noise_field: integer arithmetic alone, no library's random stream) since round 6; rounds 2-5 drew it
    # from numpy's PCG64 and pinned the bytes here.  The digests stay: they say at once when somebody changes the generator without re-cutting the fixtures.
    import hashlib
    assert hashlib.md5(b"".join(p.tobytes() for f in a for p in f)).hexdigest() == "770c659ce3b028ad7186a5df3553239f"
    assert hashlib.md5(b"".join(p.tobytes() for p in T.survey_clip(1920, 1080, 8, 2, 0, 1)[0])).hexdigest() == "191db4663f5547c0fdae4dbc00192103"
    # ... and the generator against its definition, sample by sample
    x, want = 0x1234ABCD, []
    for _ in range(5000):
        want.append((((x >> 8) * 25) >> 24) - 12)
        x = (1664525 * x + 1013904223) & 0xFFFFFFFF
    got = T.lcg_noise_field(0x1234ABCD, 50, 100).ravel().tolist()
    assert got == want and min(got) == -12 and max(got) == 12


@pytest.mark.gpu
@pytest.mark.parametrize("tag", sorted(T.FULL_CASES))
def test_encoder_object_full_size(tag):
    g = json.load(open(GOLD))[tag]
    (w, h), n, depth, cfg_id, cfg, _ = T.FULL_CASES[tag]
    stream, coded = T.encoder_run(T.load_hip(depth), T.full_case_frames(tag), w, h, **cfg)
    assert len(coded) == n
    for (poc, _, _, planes) in coded:
        got = hashlib.md5(b"".join(np.ascontiguousarray(p).tobytes() for p in planes)).hexdigest()
        assert got == g["recon_md5"][poc], "reconstruction of poc %d" % poc
    assert len(stream) == g["stream_bytes"] and hashlib.md5(stream.tobytes()).hexdigest() == g["stream_md5"]


# ---- the presets as they come (round 6): the reference's command line is `--preset <p> --no-info` and nothing else -- constant rate factor 28, aq-mode 2, cuTree ----
def test_preset_golden_present_and_cut_without_a_qp():
    g = json.load(open(PRESET_GOLD))
    for tag, ((w, h), n, depth, cfg_id, cfg, cli) in T.PRESET_CASES.items():
        assert tag in g and len(g[tag]["recon_md5"]) == n and g[tag]["stream_bytes"] > 0, tag
        assert "--qp" not in g[tag]["reference_command_line"] and "--crf" not in g[tag]["reference_command_line"] and "qp" not in cfg, tag
        assert cfg["rateControlMode"] == 2 and cfg["aqMode"] == 2 and cfg["cuTree"] == 1


@pytest.mark.gpu
@pytest.mark.parametrize("tag", sorted(T.PRESET_CASES))
def test_encoder_object_presets_as_they_come(tag):
    """BASELINE.json's configurations with their literal rate control: the whole byte stream and every reconstructed picture equal the reference command line
    encoder's for `--preset medium` (1080p 60 frames, 2160p 20 frames, Main 10), `--preset slow` and `--preset veryslow --rd 6` -- CRF's picture QPs, the adaptive
    quantisation and cuTree offsets of every 16x16 block, the QP of every CU and cu_qp_delta in the slice data"""
    g = json.load(open(PRESET_GOLD))[tag]
    (w, h), n, depth, cfg_id, cfg, _ = T.PRESET_CASES[tag]
    stream, coded = T.encoder_run(T.load_hip(depth), T.full_case_frames(tag), w, h, **cfg)
    assert len(coded) == n
    for (poc, _, _, planes) in coded:
        got = hashlib.md5(b"".join(np.ascontiguousarray(p).tobytes() for p in planes)).hexdigest()
        assert got == g["recon_md5"][poc], "reconstruction of poc %d" % poc
    assert len(stream) == g["stream_bytes"] and hashlib.md5(stream.tobytes()).hexdigest() == g["stream_md5"]


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["crf_wqvga_medium_30/", "crf_fhd_medium_60/"])
def test_device_chains_verified_under_delta_qp(tag):
    """The device-run paths of P / B pictures under the preset's rate control (delta QP: a QP per quantisation group, cu_qp_delta priced in the residual modes, the predicted
    QP of each group and topSkipMinDepth's QP test tracked on the device -- DESIGN.md section 4.27): with X265AMD_CHAIN_VERIFY=2 the host repeats every skipped CU, every
    device-made merge check and every fused search on its own path and compares mode, candidate, costs, bits, levels, QPs and the entropy coder's state.  A process of its own
    (the variable is read once); the stream must be the reference's and the paths must have run."""
    import subprocess
    import sys
    code = ("import sys, json, hashlib; sys.path.insert(0, %r); import numpy as np, hevc_testlib as T\n"
            "g = json.load(open(%r))[%r]\n"
            "(w, h), n, depth, cfg_id, cfg, _ = T.PRESET_CASES[%r]\n"
            "stream, coded = T.encoder_run(T.load_hip(depth), T.full_case_frames(%r), w, h, **cfg)\n"
            "assert len(coded) == n and hashlib.md5(stream.tobytes()).hexdigest() == g['stream_md5']\n"
            "print('verified pictures', len(coded))\n") % (os.path.dirname(os.path.abspath(__file__)), PRESET_GOLD, tag, tag, tag)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, X265AMD_CHAIN_VERIFY="2", X265AMD_TIMING="1"), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "verified pictures" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    assert "chain verify" not in r.stderr and "search verify" not in r.stderr, r.stderr[-3000:]
    import re
    runs = [int(m.group(1)) for m in re.finditer(r"skip chains so far: (\d+) commands", r.stderr)]
    ahead = [int(m.group(1)) + int(m.group(2)) for m in re.finditer(r"searches started ahead so far: (\d+) beside a leaf's merge check, (\d+) behind", r.stderr)]
    assert runs and max(runs) > 100 and ahead and max(ahead) > 0, (runs[-3:], ahead[-3:])           # the paths under test did run under delta QP


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["crf_wqvga_medium_30/", "crf_cfg4_2160p_main10/"])
def test_device_decision_of_64x64_cus_with_levels_verified(tag):
    """a 64x64 CU whose residual has a level somewhere decided on the device (the flagged units' full chains, the residual tree's rate-distortion walk:
    chain_merge_rd64) -- off by default since it gains nothing, asked for here (X265AMD_CHAIN_64=1) so that it does not rot: under X265AMD_CHAIN_VERIFY=2 every CU it skips is
    repeated by the host's own merge check (mode, candidate, cost, coder state) and every CU it hands back goes on from the device's candidate, compared the same way; the
    stream must be the reference's (8-bit at 416x240, Main 10 at 3840x2160)"""
    import subprocess
    import sys
    if tag not in T.PRESET_CASES:
        pytest.skip("no such preset case")
    code = ("import sys, json, hashlib; sys.path.insert(0, %r); import numpy as np, hevc_testlib as T\n"
            "g = json.load(open(%r))[%r]\n"
            "(w, h), n, depth, cfg_id, cfg, _ = T.PRESET_CASES[%r]\n"
            "n = min(n, 8)\n"
            "stream, coded = T.encoder_run(T.load_hip(depth), T.full_case_frames(%r)[:n], w, h, **cfg)\n"
            "print('verified pictures', len(coded))\n") % (os.path.dirname(os.path.abspath(__file__)), PRESET_GOLD, tag, tag, tag)
    env = dict(os.environ, X265AMD_CHAIN_VERIFY="2", X265AMD_CHAIN_64="1", X265AMD_TIMING="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=2400)
    assert r.returncode == 0 and "verified pictures 8" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    assert "chain verify" not in r.stderr and "search verify" not in r.stderr, r.stderr[-3000:]
    # the same eight frames without the switches: the same bytes (a clip cut short has no golden stream of its own)
    code2 = code.replace("print('verified pictures', len(coded))", "print('md5', hashlib.md5(stream.tobytes()).hexdigest())")
    a = subprocess.run([sys.executable, "-c", code2], env=dict(os.environ, X265AMD_CHAIN_64="1"), capture_output=True, text=True, timeout=2400)
    b = subprocess.run([sys.executable, "-c", code2], env=dict(os.environ), capture_output=True, text=True, timeout=2400)
    assert a.returncode == 0 and b.returncode == 0 and "md5" in a.stdout and a.stdout.strip().splitlines()[-1] == b.stdout.strip().splitlines()[-1], (a.stdout[-300:], b.stdout[-300:], a.stderr[-1500:])


def test_rc_golden_present():
    g = json.load(open(RC_GOLD))
    for tag, ((w, h), n, depth, cfg_id, cfg, cli) in T.RC_CASES.items():
        assert tag in g and len(g[tag]["recon_md5"]) == n and "--qp" not in g[tag]["reference_command_line"], tag


@pytest.mark.gpu
@pytest.mark.parametrize("tag", sorted(T.RC_CASES))
def test_encoder_object_rate_control_options(tag):
    """the rate control's own options around the preset, stream and reconstruction against the reference command line encoder's: other rate factors, cuTree off (the
    rate factor's blurred-complexity branch, fed by the AQ-weighted estimates), aq-mode 1 and 3, another strength, quantisation groups of 64, qcomp (cuTree's strength),
    no B pictures, no pyramid, short and closed GOPs, --preset slow in Main 10, rd 5 and rd 2 with delta QP"""
    g = json.load(open(RC_GOLD))[tag]
    (w, h), n, depth, cfg_id, cfg, _ = T.RC_CASES[tag]
    stream, coded = T.encoder_run(T.load_hip(depth), T.full_case_frames(tag), w, h, **cfg)
    assert len(coded) == n
    for (poc, _, _, planes) in coded:
        got = hashlib.md5(b"".join(np.ascontiguousarray(p).tobytes() for p in planes)).hexdigest()
        assert got == g["recon_md5"][poc], "reconstruction of poc %d" % poc
    assert len(stream) == g["stream_bytes"] and hashlib.md5(stream.tobytes()).hexdigest() == g["stream_md5"]
