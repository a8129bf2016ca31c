# any of the bench's configurations through the encoder object with timing / queue profile:
#   dbg/enc_cfg.py WxH medium|slow|veryslow 8|10 N [warm]   (veryslow: --preset veryslow --rd 6, BASELINE configs[4])      (X265AMD_TIMING=1 X265AMD_QUEUE_PROF=1 X265AMD_HOSTPROF=1)
import sys, os, hashlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, hevc_testlib as T, bench
w, h = (int(v) for v in sys.argv[1].split("x"))
preset, depth, N = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
Wm = int(sys.argv[5]) if len(sys.argv) > 5 else 2
bench.W, bench.H = w, h
L = T.load_hip(depth)
cfg = dict(bench.ENC_CFG, frameNumThreads=6 if h > 2000 else 5, **(T.SLOW_TOOLS if preset == "slow" else dict(T.VERYSLOW_TOOLS, **T.VERYSLOW_GOP) if preset == "veryslow" else {}))
cfg_id = {("medium", 8): 2, ("slow", 8): 3, ("medium", 10): 4, ("veryslow", 10): 5}.get((preset, depth), 2)
sync = torch.cuda.synchronize
if Wm: bench.encode(T, L, bench.bench_clip(0, Wm, depth=depth, cfg_id=cfg_id), 0, 0, sync, timed=False, cfg=cfg)
frames = bench.bench_clip(0, N, depth=depth, cfg_id=cfg_id)
bench.queue_stats(L, True)
sys.stderr.write("---- timed encode ----\n")
def _cpustat():
    try:
        t = open('/sys/fs/cgroup/cpu.stat').read().split(); return {t[i]: int(t[i + 1]) for i in range(0, len(t) - 1, 2)}
    except OSError:
        return {}
s0 = _cpustat()
c0 = os.times()
stream, dt = bench.encode(T, L, frames, 0, 0, sync, cfg=cfg)
c1 = os.times()
s1 = _cpustat()
if s0 and os.environ.get('X265AMD_TIMING'):
    sys.stderr.write('cgroup during the timed encode: %d periods, %d of them throttled, %.1f ms of thread time held back\n' % (s1['nr_periods'] - s0['nr_periods'], s1['nr_throttled'] - s0['nr_throttled'], (s1['throttled_usec'] - s0['throttled_usec']) / 1e3))
sys.stderr.write("process cpu during the timed encode: %.2f s user + %.2f s system = %.1f cores on average\n" % (c1[0] - c0[0], c1[1] - c0[1], (c1[0] - c0[0] + c1[1] - c0[1]) / dt))
print("frames", N, "seconds %.3f" % dt, "fps %.2f" % (N / dt), "bytes", len(stream), hashlib.md5(stream).hexdigest())
if os.environ.get("X265AMD_QUEUE_PROF"):
    L.lib.x265amd_queue_profile_report()
if os.environ.get("X265AMD_HOSTPROF"):
    L.lib.x265amd_hostprof_report()
