import sys, os, subprocess, tempfile, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, hevc_testlib as T
w, h, n = 192, 128, 10
planes = T.encoder_api_clip("preset_veryslow/", w, h, n, 8)
base = ["--preset", "veryslow", "--qp", "30", "--aq-mode", "0", "--no-cutree", "--no-weightp", "--no-weightb", "--b-adapt", "0", "--no-scenecut", "--keyint", "250", "--no-wpp",
        "--frame-threads", "1", "--pools", "none", "--no-info", "--no-open-gop", "--rc-lookahead", "10", "--lookahead-slices", "0", "--no-b-pyramid"]
variants = {"tu2": ["--tu-inter-depth", "2", "--tu-intra-depth", "2"], "tui": ["--tu-intra-depth", "1"], "tup": ["--tu-inter-depth", "1"], "nosh": ["--no-signhide"], "rd5": ["--rd", "5"],
            "nobintra": ["--no-b-intra"], "subme3": ["--subme", "3"], "rdoq1": ["--rdoq-level", "1"], "lm": ["--limit-modes"], "lr3": ["--limit-refs", "3"], "psyrd0": ["--psy-rd", "0"]}
out = {}
with tempfile.TemporaryDirectory() as d:
    with open(os.path.join(d, "clip.y4m"), "wb") as f:
        f.write(b"YUV4MPEG2 W%d H%d F30:1 Ip A1:1 C420\n" % (w, h))
        for fr in planes:
            f.write(b"FRAME\n")
            for pl in fr: f.write(np.ascontiguousarray(pl).tobytes())
    for name, extra in variants.items():
        r = subprocess.run([os.path.join(T.REF_DIR, "x265_ref8"), "--input", "clip.y4m", "-o", "out.hevc"] + base + extra, cwd=d, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-500:]
        out[name] = np.frombuffer(open(os.path.join(d, "out.hevc"), "rb").read(), np.uint8)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "vs_bisect.npz"), **out)
print({k: len(v) for k, v in out.items()})
