# the command sequence of a P picture that repeats the I picture (every CTU ends as a skip): X265AMD_QUEUE_TRACE=1 python dbg/enc_trace_skip.py W H
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, hevc_testlib as T
L = T.load_hip(8)
W, H = int(sys.argv[1]), int(sys.argv[2])
a = T.encoder_api_clip("trace/", W, H, 1)
planes = [a[0], a[0]]
cfg = dict(fpsNum=30, fpsDenom=1, qp=30, aspectRatioIdc=1, bframes=0, bEnableLoopFilter=1, bEnableSAO=1, bEnableWavefront=0)
stream, coded = T.encoder_run(L, planes, W, H, **cfg)
print("bytes", len(stream))
