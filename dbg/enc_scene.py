# frames 20..29 of the bench clip (the noise field is re-seeded at frame 24: poc 5 of this clip is a P picture of new content) with timing / queue logs
import sys, os, hashlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, hevc_testlib as T, bench
L = T.load_hip(8)
sync = torch.cuda.synchronize
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
frames = bench.bench_clip(20, n)
stream, dt = bench.encode(T, L, frames, 0, 0, sync)
print("frames", n, "seconds %.3f" % dt, "fps %.2f" % (n / dt), hashlib.md5(stream).hexdigest())
if os.environ.get("X265AMD_QUEUE_PROF"):
    L.lib.x265amd_queue_profile_report()
if os.environ.get("X265AMD_HOSTPROF"):
    L.lib.x265amd_hostprof_report()
