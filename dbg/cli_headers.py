"""Debugging aid: the survey clip through bin/x265amd with the given options; prints the first NAL units of the stream (sizes, types, hex of the short ones).
usage: python dbg/cli_headers.py <w> <h> <frames> [x265amd options ...]"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hevc_testlib as T
w, h, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
y4m = "/dev/shm/cli_headers.y4m"
with open(y4m, "wb") as f:
    f.write(b"YUV4MPEG2 W%d H%d F30:1 Ip A1:1 C420\n" % (w, h))
    for fr in T.survey_clip(w, h, 8, 2, 0, n):
        f.write(b"FRAME\n")
        for p in fr:
            f.write(np.ascontiguousarray(p).tobytes())
out = os.path.join(ROOT, "gpurun_out", "cli_headers.hevc")
r = subprocess.run([os.path.join(ROOT, "x265-amod_amd", "bin", "x265amd"), "--input", y4m, "-o", out] + sys.argv[4:], capture_output=True, text=True)
print(r.stderr.strip().splitlines()[-1] if r.stderr.strip() else "")
b = open(out, "rb").read()
i = 0
for k in range(10):
    j = b.find(b"\x00\x00\x00\x01", i + 4)
    if j < 0:
        j = len(b)
    print(i, j - i, (b[i + 4] >> 1) & 63, b[i:i + 48].hex() if j - i < 100 else "")
    if j >= len(b):
        break
    i = j
