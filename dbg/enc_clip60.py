# the 60-frame bench clip (scene changes at 24 and 48) through the encoder object: dbg/enc_clip60.py [frames]   (X265AMD_TIMING=1 for dbg/timeline.py)
import sys, os, hashlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, hevc_testlib as T, bench
L = T.load_hip(8)
sync = torch.cuda.synchronize
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bench.encode(T, L, bench.bench_clip(0, 4), 0, 0, sync, timed=False)
frames = bench.bench_clip(0, n)
sys.stderr.write("---- timed encode ----\n")
stream, dt = bench.encode(T, L, frames, 0, 0, sync)
print("frames", n, "seconds %.3f" % dt, "fps %.2f" % (n / dt), hashlib.md5(stream).hexdigest())
