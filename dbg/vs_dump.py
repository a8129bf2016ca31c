import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, hevc_testlib as T
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "ctudump")
os.makedirs(out, exist_ok=True)
os.environ["X265AMD_DUMP_CTU"] = out; os.environ["X265AMD_DUMP_POC"] = sys.argv[1]; os.environ["X265AMD_DUMP_MARGIN"] = "96,80"
os.environ["X265AMD_FRAME_THREADS"] = "1"
planes = T.encoder_api_clip("preset_veryslow/", 192, 128, 10, 8)
cfg = dict(fpsNum=30, fpsDenom=1, qp=30, aspectRatioIdc=1, bEnableLoopFilter=1, bEnableWavefront=0, bframes=8, bEnableSAO=1, bEnableEarlySkip=0, bEnableAMP=1, bEnableRectInter=1,
           tuQTMaxInterDepth=3, tuQTMaxIntraDepth=3, rdLevel=6, rdoqLevel=2, psyRdoqFix8=256, subpelRefine=4, maxNumMergeCand=5, searchMethod=3, maxNumReferences=5, limitReferences=0, limitModes=0)
stream, coded = T.encoder_run(T.load_hip(8), planes, 192, 128, **cfg)
print("done", len(stream), os.listdir(out))
