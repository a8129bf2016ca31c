import sys, os, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, hevc_testlib as T
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "vs_bisect.npz"))
planes = T.encoder_api_clip("preset_veryslow/", 192, 128, 10, 8)
base = dict(fpsNum=30, fpsDenom=1, qp=30, aspectRatioIdc=1, bEnableLoopFilter=1, bEnableWavefront=0, bframes=8, bEnableSAO=1, bEnableEarlySkip=0, bEnableAMP=1, bEnableRectInter=1,
            tuQTMaxInterDepth=3, tuQTMaxIntraDepth=3, rdLevel=6, rdoqLevel=2, psyRdoqFix8=256, subpelRefine=4, maxNumMergeCand=5, searchMethod=3, maxNumReferences=5, limitReferences=0, limitModes=0)
over = {"tu2": dict(tuQTMaxInterDepth=2, tuQTMaxIntraDepth=2), "tui": dict(tuQTMaxIntraDepth=1), "tup": dict(tuQTMaxInterDepth=1), "nosh": dict(bEnableSignHiding=0), "rd5": dict(rdLevel=5),
        "nobintra": dict(bIntraInBFrames=0), "subme3": dict(subpelRefine=3), "rdoq1": dict(rdoqLevel=1), "lm": dict(limitModes=1), "lr3": dict(limitReferences=3), "psyrd0": dict(psyRd=0.0)}
L = T.load_hip(8)
for name, o in over.items():
    cfg = dict(base); cfg.update(o)
    stream, coded = T.encoder_run(L, planes, 192, 128, **cfg)
    want = g[name]
    print(name, "OK" if len(stream) == len(want) and np.array_equal(stream, want) else "DIFF", len(stream), len(want))
