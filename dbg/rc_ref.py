"""Debugging aid: runs the reference encoder with its rate-control record (oracle/_ref/x265_rc_dump<D>, oracle/ref_rc_dump.cpp) on one of the test library's clips and
prints, per coded picture, what the lookahead and the rate control decided.  usage: python dbg/rc_ref.py <w> <h> <frames> [scene|survey|api] [key=value ...]"""
import os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import hevc_testlib as T

ROOT = os.path.join(os.path.dirname(__file__), "..")


def read_records(path):
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


def run(frames, w, h, depth, preset, opts, prefix):
    clip = prefix + ".yuv"
    with open(clip, "wb") as f:
        for fr in frames:
            for p in fr:
                f.write(np.ascontiguousarray(p).tobytes())
    exe = os.path.join(ROOT, "oracle", "_ref", "x265_rc_dump%d" % depth)
    subprocess.check_call([exe, clip, str(w), str(h), str(len(frames)), prefix, preset] + list(opts))
    os.remove(clip)
    return read_records(prefix + ".rc"), open(prefix + ".hevc", "rb").read()


if __name__ == "__main__":
    w, h, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    kind = sys.argv[4] if len(sys.argv) > 4 else "scene"
    opts = sys.argv[5:]
    frames = T.scene_clip(w, h, n, [n // 2]) if kind == "scene" else (T.survey_clip(w, h, 8, 2, 0, n) if kind == "survey" else T.encoder_api_clip("x", w, h, n))
    recs, stream = run(frames, w, h, 8, "medium", opts, "/tmp/rc_ref")
    print(len(stream), "bytes")
    names = {1: "IDR", 2: "I", 3: "P", 4: "Bref", 5: "B"}
    for r in recs:
        print("poc %3d %-4s ref %d sliceQp %2d qpRc %.6f qpAq %.4f satd %8d sc %d refs %s  aq[%.3f..%.3f] cutree[%.3f..%.3f] qp map %d..%d" % (
            r["poc"], names[r["type"]], r["referenced"], r["slice_qp"], r["qp_rc"], r["qp_aq"], r["satd"], r["scenecut"], r["ref_poc"],
            r["aq"].min(), r["aq"].max(), r["cutree"].min(), r["cutree"].max(), r["qp"].min(), r["qp"].max()))
