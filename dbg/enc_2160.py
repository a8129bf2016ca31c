# the bench clip at 3840x2160 through the encoder object with timing: dbg/enc_2160.py N [warm]
import sys, os, hashlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, hevc_testlib as T, bench
bench.W, bench.H = 3840, 2160
L = T.load_hip(8)
N = int(sys.argv[1]); Wm = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sync = torch.cuda.synchronize
if Wm: bench.encode(T, L, bench.bench_clip(0, Wm), 0, 0, sync, timed=False)
frames = bench.bench_clip(0, N)
sys.stderr.write("---- timed encode ----\n")
stream, dt = bench.encode(T, L, frames, 0, 0, sync)
print("frames", N, "seconds %.3f" % dt, "fps %.2f" % (N / dt), "bytes", len(stream), hashlib.md5(stream).hexdigest())
