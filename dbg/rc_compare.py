"""Debugging aid (GPU box or here for the reference half): a clip through the reference encoder with its rate-control record and through the encoder object with
X265AMD_RC_DUMP; compares picture by picture (types, QPs, block offsets, the QP / depth / mode maps) and the byte streams.
usage: python dbg/rc_compare.py <w> <h> <frames> <scene|survey|api> [ref: key=value ...] -- to cut the reference records here: add `--cut <file.npz>`; on the GPU box: `--use <file.npz>`"""
import os, pickle, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import hevc_testlib as T
import ratecontrol_lib as RL

PRESET = dict(T.FULL_BASE, aspectRatioIdc=0, rateControlMode=2, rfConstant=28.0, aqMode=2, aqStrength=1.0, cuTree=1, qCompress=0.6, qgSize=32)
NAMES = {1: "IDR", 2: "I", 3: "P", 4: "Bref", 5: "B"}


def clip(kind, w, h, n):
    if kind == "scene":
        return T.scene_clip(w, h, n, [n // 2])
    if kind == "survey":
        return T.survey_clip(w, h, 8, 2, 0, n)
    return T.encoder_api_clip("rc", w, h, n)


def main():
    a = sys.argv[1:]
    cut = use = None
    if "--cut" in a:
        i = a.index("--cut"); cut = a[i + 1]; del a[i:i + 2]
    if "--use" in a:
        i = a.index("--use"); use = a[i + 1]; del a[i:i + 2]
    w, h, n, kind = int(a[0]), int(a[1]), int(a[2]), a[3]
    opts = [o for o in a[4:] if "=" in o and not o.startswith("amd:")]
    over = {o[4:].split("=")[0]: float(o.split("=")[1]) if "." in o.split("=")[1] else int(o.split("=")[1]) for o in a[4:] if o.startswith("amd:")}
    frames = clip(kind, w, h, n)
    if use:
        recs, stream = pickle.load(open(use, "rb"))
    else:
        recs, stream = RL.reference_rc_records(frames, w, h, 8, "medium", opts, "/tmp/rc_cmp_ref")
    if cut:
        pickle.dump((recs, stream), open(cut, "wb"))
        print("reference:", len(stream), "bytes,", len(recs), "pictures ->", cut)
        return
    dump = "/tmp/rc_cmp_amd.rc"
    if os.path.exists(dump):
        os.remove(dump)
    os.environ["X265AMD_RC_DUMP"] = dump
    L = T.load_hip(8)
    got_stream, coded = T.encoder_run(L, frames, w, h, **dict(PRESET, **over))
    got = RL.read_rc_records(dump)
    print("reference %d bytes, ours %d bytes: %s" % (len(stream), len(got_stream), "IDENTICAL" if bytes(got_stream) == bytes(stream) else "differ"))
    if bytes(got_stream) != bytes(stream):
        a_, b_ = bytes(stream), bytes(got_stream)
        first = next((i for i in range(min(len(a_), len(b_))) if a_[i] != b_[i]), min(len(a_), len(b_)))
        def nals(x):
            out = []; i = 0
            while True:
                j = x.find(b"\x00\x00\x00\x01", i)
                if j < 0: break
                out.append(j); i = j + 4
            return out
        na, nb = nals(a_), nals(b_)
        print("first difference at byte", first, "; NAL starts ref", na[:8], "ours", nb[:8])
        k = max(i for i in range(len(na)) if na[i] <= first)
        print("  in NAL", k, "type", (a_[na[k] + 4] >> 1) & 63, ": ref", a_[na[k]:na[k] + 24].hex(), "ours", b_[nb[k]:nb[k] + 24].hex())
        os.makedirs("gpurun_out", exist_ok=True)
        open("gpurun_out/ours_%dx%d.hevc" % (w, h), "wb").write(b_)
    modemap = {0: 0, 1: 1, 2: 2, 3: 5}
    for k, (r, g) in enumerate(zip(recs, got)):
        notes = []
        if (r["poc"], r["type"]) != (g["poc"], g["type"]): notes.append("POC/TYPE ref %d %s ours %d %s" % (r["poc"], NAMES[r["type"]], g["poc"], NAMES[g["type"]]))
        if r["slice_qp"] != g["slice_qp"] or r["qp_rc"] != g["qp_rc"]: notes.append("QP ref %d %.9f ours %d %.9f" % (r["slice_qp"], r["qp_rc"], g["slice_qp"], g["qp_rc"]))
        if r["ref_poc"] != g["ref_poc"]: notes.append("refs %s / %s" % (r["ref_poc"], g["ref_poc"]))
        if r["scenecut"] != g["scenecut"]: notes.append("scenecut %d / %d" % (r["scenecut"], g["scenecut"]))
        for key in ("aq", "inv_qscale", "intra_cost"):
            if not np.array_equal(r[key], g[key]): notes.append("%s differs at %d blocks (max %.4g)" % (key, int((r[key] != g[key]).sum()), float(np.abs(r[key] - g[key]).max())))
        if r["referenced"] and not np.array_equal(r["cutree"].view(np.uint64), g["cutree"].view(np.uint64)):
            notes.append("cutree differs at %d of %d blocks (max %.4g)" % (int((r["cutree"] != g["cutree"]).sum()), r["cutree"].size, float(np.abs(r["cutree"] - g["cutree"]).max())))
        gm = np.vectorize(modemap.get)(g["mode"])
        for key, a_, b_ in (("qp map", r["qp"], g["qp"]), ("depth map", r["depth"], g["depth"]), ("mode map", r["mode"], gm)):
            if not np.array_equal(a_, b_):
                bad = np.argwhere(a_ != b_)
                notes.append("%s differs at %d units, first (y4 %d, x4 %d): ref %d ours %d" % (key, len(bad), bad[0][0], bad[0][1], a_[bad[0][0], bad[0][1]], b_[bad[0][0], bad[0][1]]))
        print("%2d poc %3d %-4s qp %d %.4f %s" % (k, r["poc"], NAMES[r["type"]], r["slice_qp"], r["qp_rc"], "ok" if not notes else "; ".join(notes)))


if __name__ == "__main__":
    main()
