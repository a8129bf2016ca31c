# timing of the encoder object: dbg/enc_q.py W H N [bframes]   (X265AMD_QUEUES=0 for the launch path)
import sys, os, time, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, hevc_testlib as T
L = T.load_hip(8)
W, H, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
planes = T.encoder_api_clip("big/", W, H, N)
cfg = dict(fpsNum=30, fpsDenom=1, qp=30, aspectRatioIdc=1, bframes=int(sys.argv[4]) if len(sys.argv) > 4 else 2, bEnableLoopFilter=1, bEnableSAO=1, bEnableWavefront=1)
T.encoder_run(L, planes[:2], W, H, **cfg)
t0 = time.perf_counter()
stream, coded = T.encoder_run(L, planes, W, H, **cfg)
dt = time.perf_counter() - t0
print("queues", os.environ.get("X265AMD_QUEUES", "default"), "%dx%d" % (W, H), "frames", len(coded), "seconds %.3f" % dt, "fps %.2f" % (len(coded) / dt), "bytes", len(stream),
      hashlib.md5(stream.tobytes()).hexdigest())
if os.environ.get("X265AMD_QUEUE_PROF"):
    L.lib.x265amd_queue_profile_report()
