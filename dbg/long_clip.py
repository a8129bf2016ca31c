# a long clip through the encoder object against the reference encoder (oracle/_ref) on this box: dbg/long_clip.py WxH frames [noref]
# (the lookahead's device-resident fields are trimmed beyond 2048 entries: a clip of a few hundred frames runs through that)
import sys, os, hashlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, hevc_testlib as T, bench
w, h = (int(v) for v in sys.argv[1].split("x")); n = int(sys.argv[2])
bench.W, bench.H = w, h
L = T.load_hip(8)
frames = bench.bench_clip(0, n)
stream, dt = bench.encode(T, L, frames, 0, 0, torch.cuda.synchronize)
print("x265amd: %d frames in %.2f s (%.1f fps), %d bytes, md5 %s" % (n, dt, n / dt, len(stream), hashlib.md5(stream).hexdigest()))
if len(sys.argv) > 3 and sys.argv[3] == "noref":
    sys.exit(0)
ref = bench.reference_encode(frames, runs=("default",))
print("reference: %.2f s; streams %s" % (ref["default"]["seconds"], "IDENTICAL" if ref["default"]["stream"] == stream else "DIFFER"))
