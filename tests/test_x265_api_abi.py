"""The OUTER drop-in boundary as the reference's ABI: libx265amd_main{,10}.so export x265_api_get_209 / x265_api_query returning a table with the
layout of struct x265_api (reference: source/x265.h:2561-2635, source/encoder/api.cpp:1034-1279), whose encoder entries take the reference's own x265_param /
x265_picture (x265-amod_amd/host/x265_api_abi.cpp; member offsets generated from the reference's header by oracle/gen_abi_layout.cpp).

CPU: the table, its fail-clean entries, the by-name rejections of x265_encoder_open, and every generated offset pinned against the reference's header.
GPU: oracle/_ref/x265_abi_driver{8,10} -- a libx265 CLIENT built from the reference's objects that fills x265_param with the reference's own
x265_param_default_preset / x265_param_parse and then encodes through OUR table -- writes the reference encoder's stream (golden data)."""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np
import pytest

import hevc_testlib as T

LAYOUT = {}
for line in open(os.path.join(T.PKG_DIR, "host", "x265_abi_layout.h")):
    if line.startswith("#define X265ABI_"):
        k, v = line.split()[1:3]
        LAYOUT[k[8:]] = int(v)


class Api(C.Structure):
    _fields_ = [("api_major_version", C.c_int), ("api_build_number", C.c_int), ("sizeof_param", C.c_int), ("sizeof_picture", C.c_int), ("sizeof_analysis_data", C.c_int),
                ("sizeof_zone", C.c_int), ("sizeof_stats", C.c_int), ("bit_depth", C.c_int), ("version_str", C.c_char_p), ("build_info_str", C.c_char_p),
                ("fn", C.c_void_p * 20), ("sizeof_frame_stats", C.c_int), ("fn2", C.c_void_p * 9), ("zone_param_parse", C.c_void_p)]


def table(depth):
    lib = C.CDLL(T.hip_lib_path(depth)) if hasattr(T, "hip_lib_path") else C.CDLL(os.path.join(T.PKG_DIR, "lib", "libx265amd_main.so" if depth == 8 else "libx265amd_main10.so"))
    lib.x265_api_get_209.restype = C.POINTER(Api); lib.x265_api_get_209.argtypes = [C.c_int]
    lib.x265_api_query.restype = C.POINTER(Api); lib.x265_api_query.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]
    return lib


@pytest.mark.parametrize("depth", [8, 10])
def test_api_table(depth):
    lib = table(depth)
    api = lib.x265_api_get_209(0).contents
    assert C.sizeof(Api) == LAYOUT["SIZEOF_API"]
    assert (api.api_major_version, api.api_build_number, api.bit_depth) == (LAYOUT["MAJOR_VERSION"], 209, depth)
    assert (api.sizeof_param, api.sizeof_picture, api.sizeof_stats, api.sizeof_frame_stats) == (LAYOUT["SIZEOF_PARAM"], LAYOUT["SIZEOF_PICTURE"], LAYOUT["SIZEOF_STATS"], LAYOUT["SIZEOF_FRAME_STATS"])
    assert all(api.fn[i] for i in range(20)) and all(api.fn2[i] for i in range(9)) and api.zone_param_parse
    assert not lib.x265_api_get_209(18 - depth)                 # the other depth lives in the sibling library
    err = C.c_int(-1)
    assert not lib.x265_api_query(0, 50, C.byref(err)) and err.value == 1          # X265_API_QUERY_ERR_VER_REFUSED (api.cpp:1190-1195)
    assert lib.x265_api_query(depth, 209, C.byref(err)) and err.value == 0


def _fns(lib, depth=8):
    api = lib.x265_api_get_209(depth).contents
    return dict(alloc=C.CFUNCTYPE(C.c_void_p)(api.fn[0]), free=C.CFUNCTYPE(None, C.c_void_p)(api.fn[1]), default=C.CFUNCTYPE(None, C.c_void_p)(api.fn[2]),
                parse=C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_char_p)(api.fn[3]), preset=C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_char_p)(api.fn[6]),
                open=C.CFUNCTYPE(C.c_void_p, C.c_void_p)(api.fn[10]), api=api)


def test_encoder_open_names_what_it_rejects():
    """param_default gives the reference's defaults -- CRF 28, aq-mode 2, cuTree, b-adapt 2, lookahead slices, B pyramid, the info SEI: all built since round 6 (the GPU test below
    opens and encodes with them).  What lies outside the built subset is refused BY NAME before anything touches a device"""
    lib = table(8)
    f = _fns(lib)
    lib.x265amd_last_error.restype = C.c_char_p
    p = f["alloc"]()
    f["default"](p)
    buf = (C.c_ubyte * LAYOUT["SIZEOF_PARAM"]).from_address(p)
    rd = lambda name: int.from_bytes(bytes(buf[LAYOUT["PARAM_" + name]:LAYOUT["PARAM_" + name] + 4]), "little", signed=True)
    wr = lambda name, v: buf.__setitem__(slice(LAYOUT["PARAM_" + name], LAYOUT["PARAM_" + name] + 4), list(int(v).to_bytes(4, "little", signed=True)))
    assert (rd("bframes"), rd("bFrameAdaptive"), rd("scenecutThreshold"), rd("maxNumReferences"), rd("rdLevel"), rd("searchRange"), rd("rc_rateControlMode"), rd("rc_aqMode"), rd("rc_cuTree"), rd("bEmitInfoSEI")) == (4, 2, 40, 3, 3, 57, 2, 2, 1, 1)
    wr("sourceWidth", 64); wr("sourceHeight", 64); wr("fpsNum", 30); wr("fpsDenom", 1)
    for name, value, word in (("rc_rateControlMode", 0, b"rc.rateControlMode"), ("bFrameAdaptive", 3, b"bFrameAdaptive"), ("rc_hevcAq", 1, b"hevc-aq"), ("maxCUSize", 32, b"maxCUSize"),
                              ("rc_bEnableGrain", 1, b"rc.bEnableGrain"), ("bEnableTransformSkip", 1, b"bEnableTransformSkip"), ("searchMethod", 2, b"searchMethod")):
        keep = rd(name)
        wr(name, value)
        assert not f["open"](p) and word in lib.x265amd_last_error(), (name, lib.x265amd_last_error())
        wr(name, keep)
    f["free"](p)


PRESETS = ["ultrafast", "superfast", "veryfast", "faster", "fast", "medium", "slow", "slower", "veryslow", "placebo"]
# the members x265_encoder_open reads or refuses by name (x265_api_abi.cpp) and the option tables / parser write: name -> bytes
_MEMBERS_I = ["bEnableWavefront", "frameNumThreads", "internalBitDepth", "internalCsp", "bOpenGOP", "keyframeMin", "keyframeMax", "bframes", "bFrameAdaptive", "bBPyramid", "lookaheadDepth",
              "lookaheadSlices", "scenecutThreshold", "maxCUSize", "minCUSize", "maxTUSize", "bEnableRectInter", "bEnableAMP", "tuQTMaxInterDepth", "tuQTMaxIntraDepth", "limitTU", "rdoqLevel",
              "bEnableSignHiding", "bEnableTransformSkip", "bEnableStrongIntraSmoothing", "maxNumMergeCand", "limitReferences", "limitModes", "searchMethod", "subpelRefine", "searchRange",
              "bEnableTemporalMvp", "bEnableWeightedPred", "bEnableWeightedBiPred", "bEnableLoopFilter", "bEnableSAO", "rdLevel", "bEnableEarlySkip", "recursionSkipMode", "bEnableFastIntra",
              "bIntraInBFrames", "maxNumReferences", "bEmitInfoSEI", "bAnnexB", "maxSlices", "rc_rateControlMode", "rc_qp", "rc_aqMode", "rc_cuTree", "rc_qpMin", "rc_qpMax", "rc_qgSize", "rc_hevcAq",
              "rc_qpStep", "rc_vbvBufferSize", "rc_vbvMaxBitrate", "rc_bitrate", "bLossless", "bRepeatHeaders", "levelIdc", "rc_bEnableGrain", "rc_bEnableConstVbv",
              "deblockingFilterBetaOffset", "deblockingFilterTCOffset", "bHistBasedSceneCut", "vui_aspectRatioIdc", "vui_sarWidth", "vui_sarHeight", "vui_bEnableOverscanInfoPresentFlag",
              "vui_bEnableOverscanAppropriateFlag", "vui_bEnableVideoSignalTypePresentFlag", "vui_videoFormat", "vui_bEnableVideoFullRangeFlag", "vui_bEnableColorDescriptionPresentFlag",
              "vui_colorPrimaries", "vui_transferCharacteristics", "vui_matrixCoeffs", "vui_bEnableChromaLocInfoPresentFlag", "vui_chromaSampleLocTypeTopField",
              "vui_chromaSampleLocTypeBottomField", "vui_bEnableDefaultDisplayWindowFlag", "vui_defDispWinLeftOffset", "vui_defDispWinRightOffset", "vui_defDispWinTopOffset", "vui_defDispWinBottomOffset", "bEnableAccessUnitDelimiters", "decodedPictureHashSEI",
              "bEmitHDR10SEI", "bEmitCLL", "fpsNum", "fpsDenom", "sourceWidth", "sourceHeight"]
_MEMBERS_D = ["psyRd", "psyRdoq", "rc_ipFactor", "rc_pbFactor", "rc_rfConstant", "rc_aqStrength", "rc_qCompress"]


def _members(p):
    buf = (C.c_ubyte * LAYOUT["SIZEOF_PARAM"]).from_address(p)
    out = {}
    for n in _MEMBERS_I:
        out[n] = int.from_bytes(bytes(buf[LAYOUT["PARAM_" + n]:LAYOUT["PARAM_" + n] + 4]), "little", signed=True)
    for n in _MEMBERS_D:
        out[n] = bytes(buf[LAYOUT["PARAM_" + n]:LAYOUT["PARAM_" + n] + 8]).hex()
    return out


def _reference_api():
    R = C.CDLL(T.ref_lib_path(8)) if hasattr(T, "ref_lib_path") else C.CDLL(os.path.join(T.REF_DIR, "libx265_ref8.so"))
    R.x265_param_alloc.restype = C.c_void_p
    R.x265_param_free.argtypes = [C.c_void_p]
    R.x265_param_default_preset.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    R.x265_param_parse.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    return R


@pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref (the reference build) is not present")
@pytest.mark.parametrize("preset", PRESETS + ["0", "9", "6"])
def test_param_default_preset_matches_the_references(preset):
    """our table's x265_param_default_preset against the reference library's own, for all ten presets (by name and by number) and every tune: every member that
    x265_encoder_open reads or the option tables write holds the same bytes (doubles bit for bit: rc.ipFactor is the float literal 1.4f widened)"""
    R, f = _reference_api(), _fns(table(8))
    for tune in (None, b"psnr", b"ssim", b"fastdecode", b"zero-latency", b"grain", b"animation", b"vmaf"):
        a, b = R.x265_param_alloc(), f["alloc"]()
        assert R.x265_param_default_preset(a, preset.encode(), tune) == 0 and f["preset"](b, preset.encode(), tune) == 0
        ma, mb = _members(a), _members(b)
        assert ma == mb, {k: (ma[k], mb[k]) for k in ma if ma[k] != mb[k]}
        R.x265_param_free(a); f["free"](b)
    b = f["alloc"]()
    assert f["preset"](b, b"nosuchpreset", None) == -1 and f["preset"](b, b"medium", b"nosuchtune") == -1
    f["free"](b)


PARSE_CASES = [("crf", "23.5"), ("qp", "30"), ("bframes", "3"), ("b-adapt", "1"), ("no-b-pyramid", None), ("open-gop", "0"), ("keyint", "120"), ("min-keyint", "12"), ("ref", "4"),
               ("rd", "4"), ("rdoq-level", "2"), ("psy-rd", "1.5"), ("psy-rdoq", "1.0"), ("me", "star"), ("me", "0"), ("subme", "3"), ("merange", "44"), ("max-merge", "4"),
               ("rect", None), ("amp", "1"), ("no-sao", None), ("no-deblock", None), ("no-wpp", None), ("weightb", None), ("no-weightp", None), ("aq-mode", "3"), ("aq-strength", "0.8"),
               ("no-cutree", None), ("qcomp", "0.7"), ("ipratio", "1.3"), ("pbratio", "1.2"), ("qg-size", "64"), ("rc-lookahead", "30"), ("lookahead-slices", "0"), ("scenecut", "0"),
               ("tu-intra-depth", "2"), ("tu-inter-depth", "3"), ("limit-refs", "1"), ("limit-modes", None), ("no-early-skip", None), ("rskip", "0"), ("b-intra", "0"), ("no-signhide", None),
               ("no-strong-intra-smoothing", None), ("no-temporal-mvp", None), ("fast-intra", None), ("no-info", None), ("frame-threads", "2"), ("bitrate", "1000"), ("qpmin", "10"), ("qpmax", "40"),
               ("no-scenecut", None), ("sar", "1"), ("sar", "16:11"), ("sar", "7:5"), ("sar", "x"), ("colorprim", "bt2020"), ("colorprim", "9"), ("colorprim", "nosuch"), ("transfer", "smpte2084"),
               ("colormatrix", "bt2020nc"), ("range", "full"), ("range", "limited"), ("videoformat", "ntsc"), ("chromaloc", "2"), ("overscan", "crop"), ("overscan", "show"), ("overscan", "what"),
               ("display-window", "8,4,8,4"), ("display-window", "8,4"), ("aud", None), ("hash", "2"), ("deblock", "-2:1"), ("deblock", "3,-3"), ("deblock", "2"), ("deblock", "false"), ("deblock", None), ("hdr10", None), ("no-cll", None), ("max-cll", "1000,400"), ("max-cll", "7"), ("input-res", "416x240"), ("fps", "30000/1001"), ("fps", "25"), ("fps", "60"), ("fps", "23.976"), ("fps", "12.5"), ("rd", "x"), ("nosuchoption", "1")]


@pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref (the reference build) is not present")
def test_param_parse_matches_the_references():
    """x265_param_parse through our table against the reference library's own, option by option on a fresh `medium` param: the same return code and the same members"""
    R, f = _reference_api(), _fns(table(8))
    for name, value in PARSE_CASES:
        a, b = R.x265_param_alloc(), f["alloc"]()
        R.x265_param_default_preset(a, b"medium", None); f["preset"](b, b"medium", None)
        v = value.encode() if value is not None else None
        ra, rb = R.x265_param_parse(a, name.encode(), v), f["parse"](b, name.encode(), v)
        assert ra == rb, (name, value, ra, rb)
        ma, mb = _members(a), _members(b)
        assert ma == mb, (name, value, {k: (ma[k], mb[k]) for k in ma if ma[k] != mb[k]})
        R.x265_param_free(a); f["free"](b)


@pytest.mark.gpu
@pytest.mark.parametrize("preset", ["medium", "slow"])
def test_param_default_preset_opens_and_encodes(preset):
    """x265_api.param_default_preset + a picture size must open and encode -- the reference's defaults as they come: CRF 28, aq-mode 2, cuTree, the info SEI.  The
    parameter sets are followed by the user-data SEI unit (NAL type 39); with --no-info the stream is the encoder object's own for the same settings (compared with the
    reference's in tests/test_encoder_full_size.py)"""
    lib = table(8)
    f = _fns(lib)
    api = f["api"]
    lib.x265amd_last_error.restype = C.c_char_p
    headers = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.POINTER(T.EncNal)), C.POINTER(C.c_uint32))(api.fn[14])
    encode = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.POINTER(T.EncNal)), C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p)(api.fn[15])
    close = C.CFUNCTYPE(None, C.c_void_p)(api.fn[18])
    pic_alloc = C.CFUNCTYPE(C.c_void_p)(api.fn[7]); pic_init = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)(api.fn[9])
    w, h, n = 416, 240, 12
    frames = T.survey_clip(w, h, 8, 2, 0, n)
    streams = {}
    for info in (1, 0):
        p = f["alloc"]()
        assert f["preset"](p, preset.encode(), None) == 0
        assert f["parse"](p, b"input-res", b"%dx%d" % (w, h)) == 0 and f["parse"](p, b"fps", b"30/1") == 0          # (a fraction, as a file header gives it: "30" alone is 30000 / 1000 in the VUI, param.cpp:936-941)
        if not info:
            assert f["parse"](p, b"no-info", None) == 0
        enc = f["open"](p)
        assert enc, lib.x265amd_last_error()
        nal = C.POINTER(T.EncNal)(); nnal = C.c_uint32(0)
        assert headers(enc, C.byref(nal), C.byref(nnal)) > 0
        types = [nal[i].type for i in range(nnal.value)]
        assert types == ([32, 33, 34, 39] if info else [32, 33, 34]), types
        out = bytearray()
        if info:
            sei = bytes(nal[3].payload[:nal[3].sizeBytes])
            assert b"x265amd" in sei and b"crf=28" in sei and b"cutree=1" in sei
        for i in range(nnal.value if not info else 3):
            out += bytes(nal[i].payload[:nal[i].sizeBytes])
        pic = pic_alloc(); pic_init(p, pic)
        pbuf = (C.c_ubyte * LAYOUT["SIZEOF_PICTURE"]).from_address(pic)
        coded = 0
        for t in range(n + 1):
            if t < n:
                keep = [np.ascontiguousarray(pl) for pl in frames[t]]
                for k in range(3):
                    pbuf[LAYOUT["PIC_planes"] + 8 * k:LAYOUT["PIC_planes"] + 8 * k + 8] = list(int(keep[k].ctypes.data).to_bytes(8, "little"))
                    pbuf[LAYOUT["PIC_stride"] + 4 * k:LAYOUT["PIC_stride"] + 4 * k + 4] = list(int(keep[k].strides[0]).to_bytes(4, "little"))
                pbuf[LAYOUT["PIC_pts"]:LAYOUT["PIC_pts"] + 8] = list(int(t).to_bytes(8, "little"))
                r = encode(enc, C.byref(nal), C.byref(nnal), pic, None)
                assert r >= 0, lib.x265amd_last_error()
                if r:
                    coded += 1
                    for i in range(nnal.value):
                        out += bytes(nal[i].payload[:nal[i].sizeBytes])
            else:
                while True:
                    r = encode(enc, C.byref(nal), C.byref(nnal), None, None)
                    assert r >= 0, lib.x265amd_last_error()
                    if not r:
                        break
                    coded += 1
                    for i in range(nnal.value):
                        out += bytes(nal[i].payload[:nal[i].sizeBytes])
        assert coded == n
        close(enc); f["free"](p)
        streams[info] = bytes(out)
    assert streams[1] == streams[0]            # (the SEI unit left out of the first: everything else is the same stream)
    cfg = dict(T.PRESET_BASE, aspectRatioIdc=0, **(T.SLOW_TOOLS if preset == "slow" else {}))
    own, _ = T.encoder_run(T.load_hip(8), frames, w, h, **cfg)
    assert bytes(own) == streams[0]


@pytest.mark.needs_ref
def test_generated_layout_matches_reference_header():
    if not os.path.isdir("/root/reference/source"):
        pytest.skip("the reference's header is only present in the build container")
    cmd = ["g++", "-std=gnu++11", "-fsyntax-only", "-I" + os.path.join(T.REF_DIR, "cfg"), "-I/root/reference/source", "-I" + os.path.join(T.PKG_DIR, "host"),
           os.path.join(T.ROOT, "tests", "abi_layout_check.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # and the committed header is what the generator prints today
    exe = "/tmp/x265amd_gen_abi"
    r = subprocess.run(["g++", "-std=gnu++11", "-I" + os.path.join(T.REF_DIR, "cfg"), "-I/root/reference/source", os.path.join(T.ROOT, "oracle", "gen_abi_layout.cpp"), "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert subprocess.run([exe], capture_output=True, text=True).stdout == open(os.path.join(T.PKG_DIR, "host", "x265_abi_layout.h")).read()


def _write_y4m(path, frames, w, h, depth):
    with open(path, "wb") as f:
        f.write(b"YUV4MPEG2 W%d H%d F30:1 Ip A1:1 %s\n" % (w, h, b"C420p10" if depth == 10 else b"C420"))
        for fr in frames:
            f.write(b"FRAME\n")
            for pl in fr:
                f.write(np.ascontiguousarray(pl).tobytes())


ABI_CASES = {
    # tag -> (golden file, clip, reference command line the client parses with the reference's own x265_param_parse)
    "sao_bframes/": ("frame_pipeline_golden.npz", None, ["--bframes", "2", "--rc-lookahead", "5", "--no-b-pyramid", "--sao"]),
    "hbd_b/": ("encoder_api_golden.npz", ((192, 136), 7), ["--bframes", "2", "--rc-lookahead", "5", "--no-b-pyramid", "--sao", "--rect", "--amp"]),
    "wvga/": ("encoder_api_golden.npz", ((832, 480), 5), ["--bframes", "2", "--rc-lookahead", "5", "--no-b-pyramid", "--sao", "--wpp", "--pools", "4"]),
    # open GOPs, the trellis and scene-cut detection through the table: the command line of tests/hevc_testlib.py OG_CASES as it stands
    "og_keyint_ba/": ("encoder_og_golden.npz", "og", None),
    "wp_medium/": ("encoder_wp_golden.npz", "wp", None),        # --preset medium --qp 30 as it comes (weighted prediction on: a clip whose analysis ends without weights)
    "ls_medium/": ("encoder_ls_golden.npz", "ls", None),        # 1280x720: the lookahead in slices as well
    # fades coded WITH weights through the table: P pictures (luma and chroma weights); B pictures with --weightb, Main 10, subme 4
    "wp_fade/": ("encoder_fade_golden.npz", "fade", None),
    "wp_fade_b4_hbd/": ("encoder_fade_golden.npz", "fade", None),
    "bp_deep/": ("encoder_bp_golden.npz", "bp", None),          # B pyramid + open GOPs + the trellis + a scene cut: --preset medium's GOP structure but for weighted prediction and lookahead slices
}


@pytest.mark.gpu
@pytest.mark.parametrize("tag", sorted(ABI_CASES))
def test_libx265_client_encodes_through_our_api_table(tag, tmp_path):
    gold, clip, extra = ABI_CASES[tag]
    depth = 10 if tag.startswith("hbd") or tag.endswith("_hbd/") else 8
    driver = os.path.join(T.REF_DIR, "x265_abi_driver%d" % depth)
    assert os.path.exists(driver), "oracle/build_ref.sh builds oracle/_ref/x265_abi_driver{8,10} (it travels to the GPU box with the snapshot)"
    cli = None
    if clip in ("og", "bp", "ls", "wp", "fade"):
        cases, frames_of, base = {"og": (T.OG_CASES, T.og_case_frames, T.OG_CLI), "bp": (T.BP_CASES, T.bp_case_frames, T.BP_CLI), "ls": (T.LS_CASES, T.ls_case_frames, T.LS_CLI),
                                  "wp": (T.WP_CASES, T.wp_case_frames, T.WP_CLI), "fade": (T.FADE_CASES, T.fade_case_frames, T.WP_CLI)}[clip]
        (w, h), n, depth, _, _, extra = cases[tag]
        frames = frames_of(tag)
        cli = list(base)
    elif clip is None:
        frames, stride, cstride, org = T.frame_clip_b(8)
        frames = [T.frame_planes(f, stride, cstride, org) for f in frames]
        w, h = T.MC_W, T.MC_H
    else:
        (w, h), n = clip
        frames = T.encoder_api_clip(tag, w, h, n, depth)
    want = np.load(os.path.join(T.GOLDEN_DIR, gold))[tag + "stream"]
    if cli is None:
        cli = [a for a in T.FRAME_CLI_ARGS if a not in ("--no-deblock", "--no-sao")]
    if "--wpp" in extra and "--no-wpp" in cli:
        cli = [a for a in cli if a != "--no-wpp"]
    lib = os.path.join(T.PKG_DIR, "lib", "libx265amd_main.so" if depth == 8 else "libx265amd_main10.so")
    _write_y4m(tmp_path / "clip.y4m", frames, w, h, depth)
    r = subprocess.run([driver, lib, str(tmp_path / "clip.y4m"), str(tmp_path / "out.hevc")] + cli + extra, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.fromfile(tmp_path / "out.hevc", np.uint8)
    assert len(got) == len(want) and hashlib.md5(got.tobytes()).hexdigest() == hashlib.md5(want.tobytes()).hexdigest()
    # pic_out's time stamps (pts of the picture, dts behind the B-frame reordering): what the reference library itself hands the same client
    reflib = os.path.join(T.REF_DIR, "libx265_ref%d.so" % depth)
    assert os.path.exists(reflib), "oracle/_ref/libx265_ref%d.so is not built" % depth
    rr = subprocess.run([driver, reflib, str(tmp_path / "clip.y4m"), str(tmp_path / "ref.hevc")] + cli + extra, capture_output=True, text=True, timeout=600)
    assert rr.returncode == 0, rr.stderr[-2000:]
    ours, theirs = [l for l in r.stdout.splitlines() if l.startswith("pic ")], [l for l in rr.stdout.splitlines() if l.startswith("pic ")]
    assert ours and ours == theirs, (ours[:6], theirs[:6])
