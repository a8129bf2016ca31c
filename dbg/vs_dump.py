"""usage: vs_dump.py <EDGE_CONFIGS tag> <poc | -1>: encodes the tag's clip; poc >= 0: dumps that picture's CTU states (X265AMD_DUMP_CTU) under gpurun_out/ctudump;
poc -1: only reports the first picture whose reconstruction differs from the golden data"""
import sys, os, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, hevc_testlib as T, test_encoder_api as tea
tag, poc = sys.argv[1], int(sys.argv[2])
(w, h), n, cfg = tea.EDGE_CONFIGS[tag]
depth = 10 if tag.startswith("hbd") else 8
if poc >= 0:
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "ctudump")
    os.makedirs(out, exist_ok=True)
    os.environ["X265AMD_DUMP_CTU"] = out; os.environ["X265AMD_DUMP_POC"] = str(poc); os.environ["X265AMD_DUMP_MARGIN"] = "96,80"
os.environ["X265AMD_FRAME_THREADS"] = "1"
stream, coded = T.encoder_run(T.load_hip(depth), T.encoder_api_clip(tag, w, h, n, depth), w, h, **cfg)
g = np.load(tea.EDGE_GOLD)
for (p, t, q, planes) in coded:
    ok = hashlib.md5(b"".join(np.ascontiguousarray(x).tobytes() for x in planes)).hexdigest() == str(g[tag + "recon_md5"][p])
    print("poc", p, "type", t, "qp", q, "OK" if ok else "DIFF")
