"""usage: chain_diff.py <EDGE_CONFIGS tag>: the first picture of the tag's clip with the device chain of 8x8 CUs on and off (X265AMD_INTRA_CHAIN), where the
reconstructions part"""
import sys, os, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
if len(sys.argv) > 2:
    import hevc_testlib as T, test_encoder_api as tea
    tag = sys.argv[1]
    (w, h), n, cfg = tea.EDGE_CONFIGS[tag]
    depth = 10 if tag.startswith("hbd") else 8
    cfg = dict(cfg, bEnableLoopFilter=0, bEnableSAO=0)          # the analysis does not depend on the in-loop filters: without them a differing sample is a differing decision
    stream, coded = T.encoder_run(T.load_hip(depth), T.encoder_api_clip(tag, w, h, 1, depth), w, h, **cfg)
    np.save(sys.argv[2], np.concatenate([np.ascontiguousarray(p).astype(np.int32).ravel() for p in coded[0][3]]))
    sys.exit(0)
tag = sys.argv[1]
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out")
for v in ("1", "0"):
    subprocess.run([sys.executable, __file__, tag, os.path.join(out, "chain%s.npy" % v), "x"], env=dict(os.environ, X265AMD_INTRA_CHAIN=v), check=True)
a, b = np.load(os.path.join(out, "chain1.npy")), np.load(os.path.join(out, "chain0.npy"))
import test_encoder_api as tea
(w, h), _, _ = tea.EDGE_CONFIGS[tag]
ya, yb = a[:w * h].reshape(h, w), b[:w * h].reshape(h, w)
bad = np.argwhere(ya != yb)
print("luma samples that differ:", len(bad))
if len(bad):
    blocks = sorted({(int(y) // 8 * 8, int(x) // 8 * 8) for y, x in bad})
    # in coding order: CTU row, CTU column, z-order inside
    def key(b):
        y, x = b
        z = 0
        for bit in range(3):
            z |= (((x % 64) // 8 >> bit) & 1) << (2 * bit) | (((y % 64) // 8 >> bit) & 1) << (2 * bit + 1)
        return (y // 64, x // 64, z)
    blocks.sort(key=key)
    print("first 8x8 blocks in coding order (y, x):", blocks[:6], "of", len(blocks))
    y, x = blocks[0]
    print("chain:\n", ya[y:y + 8, x:x + 8]); print("one by one:\n", yb[y:y + 8, x:x + 8])
