# the clips the ranks of a multi-GPU bench run code (bench.py --shard gops): every one must get through the weight analysis without a weight (coding with weights is not built)
import sys, os, hashlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, hevc_testlib as T, bench
L = T.load_hip(8)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for r in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
    frames = bench.bench_clip(0, K, r)          # bench.py: frames = bench_clip(0, K, gop=rank)
    try:
        stream, dt = bench.encode(T, L, frames, 0, 0, torch.cuda.synchronize)
        print("rank", r, "frames", K, "fps %.1f" % (K / dt), hashlib.md5(stream).hexdigest())
    except Exception as e:
        print("rank", r, "FAILED:", str(e)[:300])
