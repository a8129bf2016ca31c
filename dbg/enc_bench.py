# the bench clip through the encoder object with timing / queue profile: dbg/enc_bench.py N [warm]   (X265AMD_TIMING=1 X265AMD_QUEUE_PROF=1)
import sys, os, time, hashlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, hevc_testlib as T, bench
L = T.load_hip(8)
N = int(sys.argv[1]); Wm = int(sys.argv[2]) if len(sys.argv) > 2 else 2
# (frame threads: the library default, as bench.py runs it)
sync = torch.cuda.synchronize
if Wm: bench.encode(T, L, bench.bench_clip(0, Wm), 0, 0, sync, timed=False)
frames = bench.bench_clip(0, N)
sys.stderr.write("---- timed encode ----\n")
stream, dt = bench.encode(T, L, frames, 0, 0, sync)
print("frames", N, "seconds %.3f" % dt, "fps %.2f" % (N / dt), "bytes", len(stream), hashlib.md5(stream).hexdigest())
if os.environ.get("X265AMD_QUEUE_PROF"):
    L.lib.x265amd_queue_profile_report()
if os.environ.get("X265AMD_HOSTPROF"):
    L.lib.x265amd_hostprof_report()
