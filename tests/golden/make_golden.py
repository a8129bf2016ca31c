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
    make_me_golden()


ME_CONFIGS = [(T.ME_HEX, 2), (T.ME_HEX, 0), (T.ME_HEX, 1), (T.ME_HEX, 5), (T.ME_HEX, 7), (T.ME_DIA, 0), (T.ME_DIA, 2),
              (T.ME_STAR, 2), (T.ME_STAR, 4)]
ME_SCENES = ((1, (5, -3)), (2, (-17, 9)), (3, (0, 0)), (4, (33, 21)))


def make_me_golden():
    """Results of the reference's own MotionEstimate::motionEstimate (via ref_motion_estimate) and sha256 of its
    BitCost tables -> tests/golden/me_golden.npz"""
    import ctypes as C
    import hashlib
    out = {}
    for depth in (8, 10):
        ref = T.load_ref(depth)
        for method, subme in ME_CONFIGS:
            for seed, motion in ME_SCENES:
                cur, rp, stride, origin = T.me_make_planes(depth, seed, motion=motion)
                jobs = T.me_jobs(seed * 100 + method * 10 + subme, 60, motion=motion, methods=(method,), submes=(subme,))
                out["me/%d/%d/%d/%d" % (depth, method, subme, seed)] = T.me_run_host(ref, cur, rp, stride, origin, jobs)
        ref.lib.ref_mvcost_table.restype = C.POINTER(C.c_uint16)
        dig = []
        for qp in range(70):
            p = ref.lib.ref_mvcost_table(qp)
            a = np.ctypeslib.as_array(C.cast(C.addressof(p.contents) - 2 * 65536, C.POINTER(C.c_uint16)), (2 * 65536 + 1,))
            dig.append(np.frombuffer(hashlib.sha256(a.tobytes()).digest(), np.uint8))
        out["mvcost_sha256/%d" % depth] = np.stack(dig)
    np.savez_compressed(os.path.join(T.GOLDEN_DIR, "me_golden.npz"), **out)
    print("wrote me_golden.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
