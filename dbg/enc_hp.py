# host-side scopes of the encoder (X265AMD_HOSTPROF=1): dbg/enc_hp.py W H N
import sys, os, time, hashlib
os.environ.setdefault("X265AMD_HOSTPROF", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, hevc_testlib as T
L = T.load_hip(8)
W, H, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
planes = T.encoder_api_clip("big/", W, H, N)
cfg = dict(fpsNum=30, fpsDenom=1, qp=30, aspectRatioIdc=1, bframes=2, bEnableLoopFilter=1, bEnableSAO=1, bEnableWavefront=1)
T.encoder_run(L, planes[:1], W, H, **cfg)
t0 = time.perf_counter()
stream, coded = T.encoder_run(L, planes, W, H, **cfg)
print("seconds %.3f" % (time.perf_counter() - t0), hashlib.md5(stream.tobytes()).hexdigest())
L.lib.x265amd_hostprof_report()
