import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import hevc_testlib as T
L = T.load_hip(8)
for mode in (0, 1):
    for b in (4096, 65536, 1 << 20):
        print("mode", mode, "bytes", b, "stale rounds of 50:", L.lib.x265amd_queue_coherence_probe(50, mode, b))
