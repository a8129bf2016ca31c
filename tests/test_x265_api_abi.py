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


def test_encoder_open_names_what_it_rejects():
    """param_default gives the reference's defaults (CRF, b-adapt 2, lookahead slices, B pyramid ...): outside the built subset, and encoder_open says which member"""
    lib = table(8)
    api = lib.x265_api_get_209(8).contents
    alloc = C.CFUNCTYPE(C.c_void_p)(api.fn[0]); free = C.CFUNCTYPE(None, C.c_void_p)(api.fn[1]); default = C.CFUNCTYPE(None, C.c_void_p)(api.fn[2])
    preset = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_char_p)(api.fn[6]); parse = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_char_p)(api.fn[3])
    opn = C.CFUNCTYPE(C.c_void_p, C.c_void_p)(api.fn[10])
    lib.x265amd_last_error.restype = C.c_char_p
    p = alloc()
    default(p)
    buf = (C.c_ubyte * LAYOUT["SIZEOF_PARAM"]).from_address(p)
    rd = lambda name: int.from_bytes(bytes(buf[LAYOUT["PARAM_" + name]:LAYOUT["PARAM_" + name] + 4]), "little", signed=True)
    assert (rd("bframes"), rd("bFrameAdaptive"), rd("scenecutThreshold"), rd("maxNumReferences"), rd("rdLevel"), rd("searchRange"), rd("rc_rateControlMode")) == (4, 2, 40, 3, 3, 57, 2)
    assert preset(p, b"medium", None) == 0 and preset(p, b"veryslow", None) == -1 and parse(p, b"qp", b"30") == -1
    buf[LAYOUT["PARAM_sourceWidth"]] = 64; buf[LAYOUT["PARAM_sourceHeight"]] = 64; buf[LAYOUT["PARAM_fpsNum"]] = 30; buf[LAYOUT["PARAM_fpsDenom"]] = 1
    assert not opn(p) and b"rc.rateControlMode" in lib.x265amd_last_error()
    buf[LAYOUT["PARAM_rc_rateControlMode"]] = 1
    buf[LAYOUT["PARAM_bFrameAdaptive"]] = 3
    assert not opn(p) and b"bFrameAdaptive" in lib.x265amd_last_error()
    buf[LAYOUT["PARAM_bFrameAdaptive"]] = 2
    # --b-adapt 2, scene-cut detection, the lookahead in slices, the B pyramid and open GOPs (the defaults) are built
    # (weighted prediction for P pictures: the analysis is built; weighted bi-prediction is off in the preset)
    assert not opn(p) and b"bEmitInfoSEI" in lib.x265amd_last_error()
    free(p)


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
