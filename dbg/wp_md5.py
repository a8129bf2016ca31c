# dbg/wp_md5.py <tag> [runs]: the stream's digest over repeated encodes in ONE process and its reconstruction of run-to-run differences
import sys, os, hashlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, hevc_testlib as T
tag = sys.argv[1]; runs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
(w, h), n, depth, _, cfg, _ = T.WP_CASES[tag]
fr = T.wp_case_frames(tag)
for r in range(runs):
    stream, coded = T.encoder_run(T.load_hip(depth), fr, w, h, **cfg)
    print(hashlib.md5(bytes(bytearray(stream))).hexdigest(), len(stream))
