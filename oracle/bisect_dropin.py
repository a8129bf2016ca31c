#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Bisects which primitive-table slot makes the reference encoder's bitstream differ when the
table is overridden by libx265amd (oracle/ref_encode_with_table.cpp, X265AMD_SLOT_RANGE).  Run on a GPU box:
    python oracle/bisect_dropin.py 8 64 64 3 ultrafast"""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
depth = int(sys.argv[1]); args = sys.argv[2:]
exe = os.path.join(here, "_ref", "x265_dropin%d" % depth)
lib = os.path.join(here, "..", "x265-amod_amd", "lib", "libx265amd_main%s.so" % ("" if depth == 8 else "10"))

def run(rng):
    env = dict(os.environ)
    if rng: env["X265AMD_SLOT_RANGE"] = "%d:%d" % rng
    r = subprocess.run([exe, lib if rng else "none"] + args, capture_output=True, text=True, env=env, timeout=600)
    return r.stdout.strip().rsplit(" ", 1)[0]

good = run(None)
N = 2281
print("cpu:", good, flush=True)
bad_slots = []
REPEATS = int(os.environ.get("BISECT_REPEATS", "1"))      # > 1 to catch run-to-run differences (races, unstaged reads)
def search(lo, hi):
    if all(run((lo, hi)) == good for _ in range(REPEATS)): return
    if hi - lo == 1:
        bad_slots.append(lo); print("BAD slot", lo, flush=True); return
    mid = (lo + hi) // 2
    search(lo, mid); search(mid, hi)
    # a difference that needs slots from both halves shows up as no bad leaf: report the range
search(0, N)
print("bad slots:", bad_slots)
