# dbg/wp_diff.py <tag>: which pictures / NAL units of a WP_CASES encode differ from the golden data
import sys, os, hashlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, hevc_testlib as T
tag = sys.argv[1]
g = np.load(os.path.join(T.GOLDEN_DIR, "encoder_wp_golden.npz"))
(w, h), n, depth, _, cfg, _ = T.WP_CASES[tag]
for k, v in [a.split("=") for a in sys.argv[2:]]:
    cfg = dict(cfg, **{k: int(v)})
stream, coded = T.encoder_run(T.load_hip(depth), T.wp_case_frames(tag), w, h, **cfg)
for (poc, st, qp, planes) in coded:
    got = hashlib.md5(b"".join(np.ascontiguousarray(p).tobytes() for p in planes)).hexdigest()
    print("poc", poc, "type", st, "qp", qp, "recon", "ok" if got == str(g[tag + "recon_md5"][poc]) else "DIFFERS")
print(T.stream_diff(stream, g[tag + "stream"]) or "stream identical")
