"""Debugging aid: the bench clip through the encoder object with X265AMD_TIMING, under the preset's own rate control (CRF) or constant QP 30; prints the per-picture times.
usage: python dbg/preset_timing.py <frames> [cqp]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
os.environ["X265AMD_TIMING"] = "1"
import hevc_testlib as T
n = int(sys.argv[1]); cqp = len(sys.argv) > 2 and sys.argv[2] == "cqp"
cfg = dict(T.FULL_BASE, frameNumThreads=5) if cqp else dict(T.PRESET_BASE, frameNumThreads=5)
L = T.load_hip(8)
frames = T.survey_clip(1920, 1080, 8, 2, 0, n)
T.encoder_run(L, frames[:4], 1920, 1080, **cfg)
t0 = time.time()
stream, coded = T.encoder_run(L, frames, 1920, 1080, **cfg)
print("encode of %d frames: %.3f s (python loop included), %d bytes" % (n, time.time() - t0, len(stream)))
