# repeats one encode and counts distinct streams: dbg/enc_rep.py W H N repeats
import sys, os, time, hashlib, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, hevc_testlib as T
L = T.load_hip(8)
W, H, N, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
planes = T.encoder_api_clip("big/", W, H, N)
cfg = dict(fpsNum=30, fpsDenom=1, qp=30, aspectRatioIdc=1, bframes=2, bEnableLoopFilter=1, bEnableSAO=1, bEnableWavefront=1)
seen = collections.Counter()
t0 = time.perf_counter()
for r in range(R):
    stream, coded = T.encoder_run(L, planes, W, H, **cfg)
    seen[hashlib.md5(stream.tobytes()).hexdigest()[:8]] += 1
print("debug", os.environ.get("X265AMD_QUEUE_DEBUG", "0"), "queues", os.environ.get("X265AMD_QUEUES", "default"), "%dx%d x %d frames x %d runs: %.1f s" % (W, H, N, R, time.perf_counter() - t0), dict(seen))
