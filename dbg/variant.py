"""usage: variant.py make|run : veryslow-like variants against the reference CLI (make: here; run: GPU box)"""
import sys, os, subprocess, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, hevc_testlib as T
w, h, n = 192, 128, 10
planes = T.encoder_api_clip("placebo_notskip/", w, h, n, 8)
base_cli = ["--preset", "veryslow", "--qp", "30", "--aq-mode", "0", "--no-cutree", "--no-weightp", "--no-weightb", "--b-adapt", "0", "--no-scenecut", "--keyint", "250", "--no-wpp",
            "--frame-threads", "1", "--pools", "none", "--no-info", "--no-open-gop", "--rc-lookahead", "10", "--lookahead-slices", "0", "--no-b-pyramid"]
base = dict(fpsNum=30, fpsDenom=1, qp=30, aspectRatioIdc=1, bEnableLoopFilter=1, bEnableWavefront=0, bframes=8, bEnableSAO=1, bEnableEarlySkip=0, bEnableAMP=1, bEnableRectInter=1,
            tuQTMaxInterDepth=3, tuQTMaxIntraDepth=3, rdLevel=6, rdoqLevel=2, psyRdoqFix8=256, subpelRefine=4, maxNumMergeCand=5, searchMethod=3, maxNumReferences=5, limitReferences=0, limitModes=0)
V = {"tu4": (["--tu-inter-depth", "4", "--tu-intra-depth", "4"], dict(tuQTMaxInterDepth=4, tuQTMaxIntraDepth=4)),
     "tu4inter": (["--tu-inter-depth", "4"], dict(tuQTMaxInterDepth=4)),
     "tu4intra": (["--tu-intra-depth", "4"], dict(tuQTMaxIntraDepth=4)),
     "tu4norq": (["--tu-inter-depth", "4", "--tu-intra-depth", "4", "--rdoq-level", "0"], dict(tuQTMaxInterDepth=4, tuQTMaxIntraDepth=4, rdoqLevel=0, psyRdoqFix8=0)),
     "other": (["--subme", "5", "--merange", "92", "--rskip", "0"], dict(subpelRefine=5, searchRange=92, recursionSkipMode=0))}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "variant.npz")
if sys.argv[1] == "make":
    out = {}
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "clip.y4m"), "wb") as f:
            f.write(b"YUV4MPEG2 W%d H%d F30:1 Ip A1:1 C420\n" % (w, h))
            for fr in planes:
                f.write(b"FRAME\n")
                for pl in fr: f.write(np.ascontiguousarray(pl).tobytes())
        for name, (extra, _) in V.items():
            r = subprocess.run([os.path.join(T.REF_DIR, "x265_ref8"), "--input", "clip.y4m", "-o", "out.hevc"] + base_cli + extra, cwd=d, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-500:]
            out[name] = np.frombuffer(open(os.path.join(d, "out.hevc"), "rb").read(), np.uint8)
    np.savez_compressed(path, **out)
else:
    g = np.load(path)
    L = T.load_hip(8)
    for name, (_, o) in V.items():
        cfg = dict(base); cfg.update(o)
        stream, coded = T.encoder_run(L, planes, w, h, **cfg)
        print(name, "OK" if len(stream) == len(g[name]) and np.array_equal(stream, g[name]) else "DIFF")
