import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, hevc_testlib as T
L = T.load_hip(8)
frames, stride, cstride, org = T.frame_clip_b(8)
planes = [T.frame_planes(f, stride, cstride, org) for f in frames]
cfg = dict(fpsNum=30, fpsDenom=1, qp=30, aspectRatioIdc=1, bframes=2, bEnableLoopFilter=1, bEnableSAO=1, bEnableWavefront=0)
T.encoder_run(L, planes, T.MC_W, T.MC_H, **cfg)
t0 = time.perf_counter()
stream, coded = T.encoder_run(L, planes, T.MC_W, T.MC_H, **cfg)
print("seconds", time.perf_counter() - t0, "frames", len(coded), "bytes", len(stream))
