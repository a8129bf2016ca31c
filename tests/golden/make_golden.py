#!/usr/bin/env python3
"""Generate the golden vectors for the hot-path primitives FROM THE REFERENCE ITSELF.

Runs every case of tests/hevc_testlib.py through oracle/_ref/librefprims{8,10}.so (the reference's own C primitive
table, built from /root/reference by oracle/build_ref.sh) and writes
  tests/golden/prims_digests.json   sha256 of the outputs of every (case, depth, mode, rep)
  tests/golden/prims_<depth>.npz    full output arrays for (case, "random", rep 0)
Inputs are not stored: they are regenerated from the per-case seed (hevc_testlib.case_seed).
Only runs in the build container (needs oracle/_ref); the outputs are committed.
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hevc_testlib as T

REPS = {"random": 3, "min": 1, "max": 1}


def main():
    digests = {}
    for depth in (8, 10):
        ref = T.load_ref(depth)
        full = {}
        for name in sorted(T.CASES):
            for mode in T.MODES:
                for rep in range(REPS[mode]):
                    outs = T.run_case(ref, name, mode, rep)
                    digests["%s/%d/%s/%d" % (name, depth, mode, rep)] = T.digest(outs)
                    if mode == "random" and rep == 0:
                        for i, a in enumerate(outs):
                            full["%s/%04d" % (name, i)] = a
        np.savez_compressed(os.path.join(T.GOLDEN_DIR, "prims_%d.npz" % depth), **full)
    with open(os.path.join(T.GOLDEN_DIR, "prims_digests.json"), "w") as f:
        json.dump({"reference": "DJATOM/x265-aMod 3.6+1-aa7f602f7 [noasm] C primitives", "digests": digests}, f, indent=0, sort_keys=True)
    print("wrote", len(digests), "digests")


if __name__ == "__main__":
    main()
