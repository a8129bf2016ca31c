# The I picture of the bench clip through the encoder object, for the counter evidence of the job server's commands (profiles/collect_traffic.sh):
#   dbg/enc_traffic.py N out.json        with the job server: per command kind its count, body time and algorithmic bytes (x265amd_queue_stats) -> out.json
#   X265AMD_QUEUES=0 dbg/enc_traffic.py N   the same encode with every command as an ordinary kernel launch (what rocprofv3 --pmc can count); prints the stream's digest
import sys, os, json, hashlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, hevc_testlib as T, bench
L = T.load_hip(8)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
frames = bench.bench_clip(0, N)
queues = os.environ.get("X265AMD_QUEUES") != "0"
if queues:
    bench.encode(T, L, frames, 0, 0, torch.cuda.synchronize, timed=False)       # warm
    bench.queue_stats(L, True)
stream, dt = bench.encode(T, L, frames, 0, 0, torch.cuda.synchronize)
print("frames", N, "seconds %.3f" % dt, "bytes", len(stream), hashlib.md5(stream).hexdigest())
if queues and len(sys.argv) > 2:
    st = bench.queue_stats(L, False)
    per_op = {bench.XA_OPS[k]: {"commands": st[10 + 3 * k], "body_ms": st[11 + 3 * k] / 1e5, "algorithmic_bytes": st[12 + 3 * k]} for k in range(len(bench.XA_OPS)) if st[10 + 3 * k]}
    json.dump({"frames": N, "stream_md5": hashlib.md5(stream).hexdigest(), "per_command_kind": per_op}, open(sys.argv[2], "w"), indent=1)
