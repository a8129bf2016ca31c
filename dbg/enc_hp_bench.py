# host-side scopes over the bench clip (X265AMD_HOSTPROF=1 or =wall): dbg/enc_hp_bench.py N
import sys, os
os.environ.setdefault("X265AMD_HOSTPROF", "1")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, hevc_testlib as T, bench
L = T.load_hip(8)
N = int(sys.argv[1])
stream, dt = bench.encode(T, L, bench.bench_clip(0, N), 0, 0, torch.cuda.synchronize)
print("frames", N, "seconds %.3f" % dt)
L.lib.x265amd_hostprof_report()
