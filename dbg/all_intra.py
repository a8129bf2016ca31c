import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch, hevc_testlib as T, bench
L = T.load_hip(8)
frames = bench.bench_clip(0, 24)
bench.encode(T, L, frames[:2], 0, 1, torch.cuda.synchronize, timed=False)
for ft in (1, 2, 3, 5):
    cfg = dict(bench.ENC_CFG, frameNumThreads=ft)
    s, dt = bench.encode(T, L, frames, 0, 1, torch.cuda.synchronize, cfg=cfg)
    print("all-intra 1080p, 24 frames, frame threads %d: %.2f s (%.1f fps)" % (ft, dt, 24 / dt))
