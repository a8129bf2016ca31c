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
def throttled():
    try: return dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat")).get("nr_throttled", "?") + "/" + dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat")).get("throttled_usec", "?")
    except Exception: return "?"
th0 = throttled()
c0 = os.times()
stream, dt = bench.encode(T, L, frames, 0, 0, sync)
c1 = os.times()
sys.stderr.write("cgroup nr_throttled/throttled_usec before the timed encode %s, after %s\n" % (th0, throttled()))
sys.stderr.write("process cpu during the timed encode: %.2f s user + %.2f s system = %.1f cores on average\n" % (c1[0] - c0[0], c1[1] - c0[1], (c1[0] - c0[0] + c1[1] - c0[1]) / dt))
print("frames", N, "seconds %.3f" % dt, "fps %.2f" % (N / dt), "bytes", len(stream), hashlib.md5(stream).hexdigest())
if os.environ.get("X265AMD_QUEUE_PROF"):
    L.lib.x265amd_queue_profile_report()
if os.environ.get("X265AMD_HOSTPROF"):
    L.lib.x265amd_hostprof_report()
if os.environ.get("THREAD_CPU"):
    # CPU seconds per thread of this process since it started (utime + stime of /proc/self/task/*/stat), busiest first
    tck = os.sysconf("SC_CLK_TCK"); rows = []
    for t in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % t).read(); name = f[f.index("(") + 1:f.rindex(")")]; v = f[f.rindex(")") + 2:].split()
            rows.append(((int(v[11]) + int(v[12])) / tck, name, t))
        except Exception: pass
    rows.sort(reverse=True)
    print("threads alive %d, cpu s of the busiest: %s" % (len(rows), " ".join("%s:%.2f" % (n, c) for c, n, _ in rows[:40])))
