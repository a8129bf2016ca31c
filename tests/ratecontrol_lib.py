"""Test plumbing for the rate-control host code (include/x265amd_ratecontrol.h): seeded cuTree cases run through the product's x265amd_cutree and through the
reference's own Lookahead::cuTree (oracle/refprims.cpp: ref_cutree); records of whole reference encodes (oracle/ref_rc_dump.cpp) replayed through x265amd_rc_start and
x265amd_cu_qp.  No GPU work anywhere in this file."""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

import hevc_testlib as T

TYPE_IDR, TYPE_I, TYPE_P, TYPE_BREF, TYPE_B = 1, 2, 3, 4, 5


class CutreeFrame(C.Structure):
    _fields_ = [("slice_type", C.c_int32), ("reserved", C.c_int32), ("intra_cost", C.c_void_p), ("inv_qscale", C.c_void_p), ("qp_aq_offset", C.c_void_p),
                ("qp_cutree_offset", C.c_void_p), ("propagate_cost", C.c_void_p), ("weighted_cost_delta", C.c_void_p)]


class CutreeParams(C.Structure):
    _fields_ = [("width8", C.c_int32), ("height8", C.c_int32), ("fps_num", C.c_uint32), ("fps_denom", C.c_uint32), ("b_pyramid", C.c_int32), ("weighted_bipred", C.c_int32),
                ("lookahead_depth", C.c_int32), ("reserved", C.c_int32), ("strength", C.c_double)]


class RcParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("fps_num", C.c_uint32), ("fps_denom", C.c_uint32), ("bframes", C.c_int32), ("keyframe_max", C.c_int32),
                ("cu_tree", C.c_int32), ("qp_min", C.c_int32), ("qp_max", C.c_int32), ("reserved", C.c_int32), ("rf_constant", C.c_double), ("q_compress", C.c_double),
                ("ip_factor", C.c_double), ("pb_factor", C.c_double)]


class RcFrame(C.Structure):
    _fields_ = [("slice_type", C.c_int32), ("is_referenced", C.c_int32), ("poc", C.c_int32), ("scenecut", C.c_int32), ("ref0_scenecut", C.c_int32), ("last_minigop_b", C.c_int32),
                ("satd_cost", C.c_int64), ("ref_slice_type", C.c_int32 * 2), ("ref_poc", C.c_int32 * 2), ("ref_is_referenced", C.c_int32 * 2), ("ref_avg_qp_rc", C.c_double * 2)]


ESTIMATE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p))


def minigop_types(rng, numframes, bframes, pyramid):
    """slice types of frames[0 .. numframes]: frames[0] a P picture, then mini-GOPs of 0..bframes B pictures ending in a P picture (the window may end in B pictures)"""
    types = [TYPE_P]
    while len(types) <= numframes:
        nb = int(rng.integers(0, bframes + 1))
        types += [TYPE_B] * nb + [TYPE_P]
    types = types[:numframes + 1]
    return types


def cutree_estimates(types, numframes, b_intra, pyramid):
    """the (p0, p1, b) triples Lookahead::cuTree asks for (slicetype.cpp:3443-3488), in its order"""
    out = []
    idx = 0 if b_intra else 1
    i = numframes
    while i > 0 and types[i] == TYPE_B:
        i -= 1
    lastnonb = i
    if lastnonb < idx:
        return out
    while True:
        i -= 1
        if not (i + 1 > idx):
            break
        curnonb = i
        while types[curnonb] == TYPE_B and curnonb > 0:
            curnonb -= 1
        if curnonb < idx:
            break
        out.append((curnonb, lastnonb, lastnonb))
        bframes = lastnonb - curnonb - 1
        if pyramid and bframes > 1:
            middle = (bframes + 1) // 2 + curnonb
            out.append((curnonb, lastnonb, middle))
            while i > curnonb:
                p0 = middle if i > middle else curnonb
                p1 = middle if i < middle else lastnonb
                if i != middle:
                    out.append((p0, p1, i))
                i -= 1
        else:
            while i > curnonb:
                out.append((curnonb, lastnonb, i))
                i -= 1
        lastnonb = curnonb
    return out


def cutree_case(seed, width=416, height=240, bframes=4, pyramid=1, weightb=0, numframes=12, b_intra=0, qcompress=0.6, fps=(30, 1), lookahead=20):
    rng = np.random.default_rng(seed)
    w8, h8 = ((width // 2) + 7) >> 3, ((height // 2) + 7) >> 3
    ncu = w8 * h8
    n = numframes + 1
    types = minigop_types(rng, numframes, bframes, pyramid)
    if b_intra:
        types[0] = TYPE_I
    c = dict(width=width, height=height, bframes=bframes, pyramid=pyramid, weightb=weightb, numframes=numframes, b_intra=b_intra, qcompress=qcompress, fps=fps, lookahead=lookahead,
             w8=w8, h8=h8, ncu=ncu, types=np.array(types, np.int32))
    intra = rng.integers(0, 6000, (n, ncu)).astype(np.int32)
    intra[rng.random((n, ncu)) < 0.03] = 0
    c["intra_cost"] = intra
    aq = rng.normal(0, 1.5, (n, ncu))
    c["qp_aq"] = aq
    c["inv_qscale"] = np.clip(np.round(256 * 2.0 ** (-aq / 6)), 1, 65535).astype(np.int32)
    c["qp_cutree"] = aq + rng.normal(0, 0.5, (n, ncu))
    c["propagate"] = rng.integers(0, 65536, (n, ncu)).astype(np.uint16)
    wcd = np.zeros((n, 18))
    mask = rng.random((n, 18)) < 0.2
    wcd[mask] = rng.uniform(0.5, 0.998, int(mask.sum()))
    c["wcd"] = wcd
    est = cutree_estimates(types, numframes, b_intra, pyramid)
    c["est"] = np.array(est, np.int32).reshape(-1, 3)
    ne = len(est)
    costs = np.zeros((ne, ncu), np.uint16)
    mv0 = np.zeros((ne, ncu, 2), np.int16); mv1 = np.zeros((ne, ncu, 2), np.int16)
    for e, (p0, p1, b) in enumerate(est):
        inter = np.minimum(rng.integers(0, 7000, ncu), 16383).astype(np.uint16)
        lists = rng.integers(1, 4, ncu) if p1 > b else np.ones(ncu, np.int64)
        lists = np.where(inter >= intra[b], 0, lists)           # an intra block uses no list (estimateCUCost: listsUsed stays 0)
        lists = np.where(rng.random(ncu) < 0.05, 0, lists)
        costs[e] = inter | (lists.astype(np.uint16) << 14)
        for mv in (mv0, mv1):
            v = rng.integers(-160, 161, (ncu, 2)).astype(np.int16)
            v[rng.random(ncu) < 0.3] = 0
            far = rng.random(ncu) < 0.02
            v[far] = rng.integers(-2000, 2001, (int(far.sum()), 2)).astype(np.int16)
            mv[e] = v
    c["lowres_costs"], c["mvs0"], c["mvs1"] = costs, mv0, mv1
    return c


def cutree_run_ref(R, c):
    tree = c["qp_cutree"].copy(); prop = c["propagate"].copy()
    ne = len(c["est"])
    recalc = np.zeros(max(ne, 1), np.int64)
    R.lib.ref_cutree.restype = C.c_int
    got = R.lib.ref_cutree(c["width"], c["height"], c["bframes"], c["pyramid"], c["weightb"], c["lookahead"], C.c_double(c["qcompress"]), c["fps"][0], c["fps"][1], c["numframes"],
                           c["b_intra"], T._ptr(c["types"]), T._ptr(c["intra_cost"]), T._ptr(c["inv_qscale"]), T._ptr(c["qp_aq"]), T._ptr(tree), T._ptr(prop), T._ptr(c["wcd"]),
                           ne, T._ptr(c["est"]), T._ptr(c["lowres_costs"]), T._ptr(c["mvs0"]), T._ptr(c["mvs1"]), T._ptr(recalc))
    assert got == c["ncu"], got
    return tree, prop, recalc[:ne]


def cutree_params(c):
    return CutreeParams(c["w8"], c["h8"], c["fps"][0], c["fps"][1], c["pyramid"], c["weightb"], c["lookahead"], 0, 5.0 * (1.0 - c["qcompress"]))


def cutree_run_prod(L, c):
    lib = L.lib
    n = c["numframes"] + 1
    tree = c["qp_cutree"].copy(); prop = c["propagate"].copy()
    frames = (CutreeFrame * n)()
    ptrs = (C.POINTER(CutreeFrame) * n)()
    for k in range(n):
        f = frames[k]
        f.slice_type = int(c["types"][k])
        f.intra_cost = c["intra_cost"][k].ctypes.data; f.inv_qscale = c["inv_qscale"][k].ctypes.data; f.qp_aq_offset = c["qp_aq"][k].ctypes.data
        f.qp_cutree_offset = tree[k].ctypes.data; f.propagate_cost = prop[k].ctypes.data; f.weighted_cost_delta = c["wcd"][k].ctypes.data
        ptrs[k] = C.pointer(f)
    index = {tuple(int(v) for v in e): i for i, e in enumerate(c["est"])}
    asked = []

    def estimate(ctx, p0, p1, b, lc, m0, m1):
        asked.append((p0, p1, b))
        e = index.get((p0, p1, b))
        if e is None:
            return -7
        lc[0] = c["lowres_costs"][e].ctypes.data
        m0[0] = c["mvs0"][e].ctypes.data if b > p0 else None
        m1[0] = c["mvs1"][e].ctypes.data if p1 > b else None
        return 0

    cb = ESTIMATE_FN(estimate)
    prm = cutree_params(c)
    rc = lib.x265amd_cutree(C.byref(prm), ptrs, c["numframes"], c["b_intra"], cb, None)
    assert rc == 0, rc
    # the first time each estimate is asked for, in order: what the reference's singleCost calls are
    first = []
    for a in asked:
        if a not in first:
            first.append(a)
    assert first == [tuple(int(v) for v in e) for e in c["est"]], (first, c["est"].tolist())
    lib.x265amd_frame_cost_recalculate.restype = C.c_int64
    recalc = np.array([-1 if c["types"][b] == TYPE_B else lib.x265amd_frame_cost_recalculate(C.byref(prm), T._ptr(c["lowres_costs"][e]), T._ptr(tree[b]))
                       for e, (p0, p1, b) in enumerate(c["est"])], np.int64)
    return tree, prop, recalc


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


# (seed, keyword overrides): pyramid and plain mini-GOPs, weighted bi-prediction's weights, the keyframe pass (b_intra), windows that end in B pictures, odd sizes, other rates
CUTREE_CASES = [(1, {}), (2, dict(pyramid=0)), (3, dict(weightb=1)), (4, dict(b_intra=1)), (5, dict(numframes=20, bframes=8, lookahead=40)), (6, dict(width=328, height=248, numframes=7)),
                (7, dict(fps=(24000, 1001), qcompress=0.7)), (8, dict(numframes=1)), (9, dict(numframes=0, b_intra=1)), (10, dict(numframes=3, bframes=3)),
                (11, dict(fps=(120, 1))), (12, dict(width=1920, height=1080, numframes=6))]


# ---- records of whole reference encodes (oracle/_ref/x265_rc_dump<depth>) ----
def read_rc_records(path):
    raw = open(path, "rb").read()
    at, out = 0, []
    while at < len(raw):
        hdr = np.frombuffer(raw, "<i4", 44, at); at += 176
        assert hdr[0] == 0x52434450
        w4, h4, b16, lb = (int(v) for v in hdr[40:44])
        r = dict(poc=int(hdr[1]), type=int(hdr[2]), referenced=int(hdr[3]), slice_qp=int(hdr[4]), scenecut=int(hdr[5]), num_ref=[int(hdr[6]), int(hdr[7])],
                 ref_poc=[hdr[8:8 + hdr[6]].tolist(), hdr[24:24 + hdr[7]].tolist()], w4=w4, h4=h4)
        r["satd"] = int(np.frombuffer(raw, "<i8", 1, at)[0]); at += 8
        r["qp_rc"], r["qp_aq"] = (float(v) for v in np.frombuffer(raw, "<f8", 2, at)); at += 16
        r["aq"] = np.frombuffer(raw, "<f8", b16, at).copy(); at += 8 * b16
        r["cutree"] = np.frombuffer(raw, "<f8", b16, at).copy(); at += 8 * b16
        r["inv_qscale"] = np.frombuffer(raw, "<i4", b16, at).copy(); at += 4 * b16
        r["intra_cost"] = np.frombuffer(raw, "<i4", lb, at).copy(); at += 4 * lb
        r["propagate"] = np.frombuffer(raw, "<u2", lb, at).copy(); at += 2 * lb
        for k in ("qp", "depth", "mode", "cbf"):
            r[k] = np.frombuffer(raw, "i1" if k == "qp" else "u1", w4 * h4, at).reshape(h4, w4).copy(); at += w4 * h4
        out.append(r)
    return out


def reference_rc_records(frames, w, h, depth, preset, opts, prefix):
    """runs the reference encoder with its per-picture record on `frames` ((Y, U, V) arrays in display order); returns (records, byte stream)"""
    clip = prefix + ".yuv"
    with open(clip, "wb") as f:
        for fr in frames:
            for p in fr:
                f.write(np.ascontiguousarray(p).tobytes())
    exe = os.path.join(T.REF_DIR, "x265_rc_dump%d" % depth)
    subprocess.check_call([exe, clip, str(w), str(h), str(len(frames)), prefix, preset] + list(opts), stderr=subprocess.DEVNULL)
    os.remove(clip)
    recs, stream = read_rc_records(prefix + ".rc"), open(prefix + ".hevc", "rb").read()
    os.remove(prefix + ".rc"); os.remove(prefix + ".hevc")
    return recs, stream


SLICE_OF = {TYPE_IDR: 2, TYPE_I: 2, TYPE_P: 1, TYPE_BREF: 0, TYPE_B: 0}
IP_FACTOR, PB_FACTOR = float(np.float32(1.4)), float(np.float32(1.3))       # param.rc.ipFactor / pbFactor are set from float literals (common/param.cpp:276-277)


def rc_replay(L, recs, w, h, bframes=4, keyint=250, rf=28.0, qcomp=0.6, cutree=1):
    """x265amd_rc_start over the pictures of a record list in coding order; returns [(slice qp, avg_qp_rc)]"""
    lib = L.lib
    lib.x265amd_rc_open.restype = C.c_void_p
    p = RcParams(w, h, 30, 1, bframes, keyint, cutree, 0, 69, 0, rf, qcomp, IP_FACTOR, PB_FACTOR)
    rc = C.c_void_p(lib.x265amd_rc_open(C.byref(p)))
    assert rc
    by, out = {}, []
    for r in recs:
        st = SLICE_OF[r["type"]]
        f = RcFrame()
        f.slice_type, f.is_referenced, f.poc, f.scenecut, f.satd_cost = st, r["referenced"], r["poc"], r["scenecut"], r["satd"]
        if st != 2:
            f.ref0_scenecut = by[r["ref_poc"][0][0]]["scenecut"]
        if st == 0:
            for l in range(2):
                q = by[r["ref_poc"][l][0]]
                f.ref_slice_type[l], f.ref_poc[l], f.ref_is_referenced[l], f.ref_avg_qp_rc[l] = SLICE_OF[q["type"]], q["poc"], q["referenced"], out[q["at"]][1]
        avg = C.c_double(0)
        qp = lib.x265amd_rc_start(rc, C.byref(f), C.byref(avg))
        r["at"] = len(out)
        out.append((qp, avg.value))
        by[r["poc"]] = r
    lib.x265amd_rc_close(rc)
    return out
