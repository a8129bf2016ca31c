"""usage: enc_sc.py <SC_CASES tag> [runs]: the scene-cut clips through the encoder object, `runs` times: frame types, per-picture reconstruction against the golden data,
where the stream parts from the golden stream; the stream is left under gpurun_out/"""
import sys, os, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, hevc_testlib as T
tag = sys.argv[1]; runs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
(w, h), n, depth, _, cfg, _ = T.SC_CASES[tag]
g = np.load(os.path.join(T.GOLDEN_DIR, "encoder_sc_golden.npz"))
for r in range(runs):
    stream, coded = T.encoder_run(T.load_hip(depth), T.scene_case_frames(tag), w, h, **cfg)
    line = []
    for (p, t, q, planes) in coded:
        ok = hashlib.md5(b"".join(np.ascontiguousarray(x).tobytes() for x in planes)).hexdigest() == str(g[tag + "recon_md5"][p])
        line.append("%d:%d:q%d:%s" % (p, t, q, "ok" if ok else "DIFF"))
    print(tag, "run", r, " ".join(line))
    print("   ", T.stream_diff(stream, g[tag + "stream"]) or "stream identical")
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "%s_run%d.hevc" % (tag.strip("/"), r))
    open(out, "wb").write(bytes(bytearray(stream)))
