# round trip of a device job queue: python dbg/q_rtt.py
import sys, os, ctypes as C
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hevc_testlib as T
L = T.load_hip(8)
L.lib.x265amd_queue_rtt_ns.restype = C.c_double
L.lib.x265amd_queue_rtt_ns.argtypes = [C.c_int, C.c_int]
for mode, what in ((0, "empty signalling command"), (1, "fill + wait (2 commands)"), (2, "3 fills + wait (4 commands)")):
    print("%-32s %.2f us" % (what, L.lib.x265amd_queue_rtt_ns(20000, mode) / 1000.0))
