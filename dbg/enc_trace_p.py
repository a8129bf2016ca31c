# the command sequence of a small P picture with new content: X265AMD_QUEUE_TRACE=1 python dbg/enc_trace_p.py W H
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, hevc_testlib as T
L = T.load_hip(8)
W, H = int(sys.argv[1]), int(sys.argv[2])
a = T.encoder_api_clip("trace/", W, H, 2)
b = T.encoder_api_clip("trace2/", W + 64, H, 2)          # other content
planes = [a[0], [pl[:H if i == 0 else H // 2, :W if i == 0 else W // 2].copy() for i, pl in enumerate(b[1])]]
half = [planes[1][0].copy(), planes[1][1].copy(), planes[1][2].copy()]
half[0][:, :W // 2] = a[1][0][:, :W // 2]; half[1][:, :W // 4] = a[1][1][:, :W // 4]; half[2][:, :W // 4] = a[1][2][:, :W // 4]      # left half moves, right half is new
planes[1] = half
cfg = dict(fpsNum=30, fpsDenom=1, qp=30, aspectRatioIdc=1, bframes=0, bEnableLoopFilter=1, bEnableSAO=1, bEnableWavefront=0)
sys.stderr.write("==== encode ====\n")
stream, coded = T.encoder_run(L, planes, W, H, **cfg)
print("bytes", len(stream))
