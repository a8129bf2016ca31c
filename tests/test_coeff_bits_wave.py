"""The wavefront form of the bit count of a 4x4 unit (x265amd_coeff_bits_wave: one lane per context, csrc/entropy_dev.h) against the oracle's
Entropy::codeCoeffNxN in counting mode (reference: source/encoder/entropy.cpp:1828-2199): bits and adapted contexts, bit for bit."""
import ctypes as C
import numpy as np
import pytest
import hevc_testlib as T

pytestmark = pytest.mark.gpu


def levels_4x4(rng, kind):
    """level patterns that reach every branch: sparse / dense, small / escape-coded magnitudes, more than eight levels, a lone DC"""
    lv = np.zeros(16, np.int64)
    if kind == 0:
        lv[rng.integers(0, 16)] = rng.integers(1, 4) * rng.choice([-1, 1])
    elif kind == 1:
        k = rng.integers(1, 6)
        idx = rng.choice(16, k, replace=False)
        lv[idx] = rng.integers(1, 3, k) * rng.choice([-1, 1], k)
    elif kind == 2:
        k = rng.integers(6, 17)
        idx = rng.choice(16, k, replace=False)
        lv[idx] = rng.integers(1, 6, k) * rng.choice([-1, 1], k)
    elif kind == 3:
        k = rng.integers(9, 17)
        idx = rng.choice(16, k, replace=False)
        lv[idx] = np.where(rng.random(k) < 0.7, 1, rng.integers(1, 400, k)) * rng.choice([-1, 1], k)
    else:
        k = rng.integers(1, 17)
        idx = rng.choice(16, k, replace=False)
        lv[idx] = rng.integers(1, 3000, k) * rng.choice([-1, 1], k)
    return lv.astype(np.int16)


@pytest.mark.parametrize("depth", [8, 10])
def test_wave_coeff_bits_4x4(depth):
    import torch
    hip, orc = T.load_hip(depth), T.load_oracle(depth)
    rng = np.random.default_rng(4400 + depth)
    n = 3000
    cases, levels = [], []
    for i in range(n):
        intra = int(rng.random() < 0.8)
        c = dict(log2=2, ttype=int(rng.integers(0, 3)), intra=intra, dir=int(rng.integers(0, 35)), signhide=int(rng.integers(0, 2)),
                 ctx=T.entropy_reset(orc, int(rng.integers(0, 3)), int(rng.integers(10, 46))) if i % 3 else rng.integers(0, 126, T.CTX_COUNT).astype(np.uint8))
        lv = levels_4x4(rng, i % 5)
        cases.append(c); levels.append((int(np.count_nonzero(lv)), lv))
    want = T.coeff_bits_run(orc, cases, levels)
    lev = np.stack([l[1] for l in levels])
    ctxs = np.zeros((n, 160), np.uint8)
    for i, c in enumerate(cases):
        ctxs[i, :T.CTX_COUNT] = c["ctx"]
    d_lev = torch.from_numpy(lev).cuda(); d_ctx = torch.from_numpy(ctxs).cuda(); d_out = torch.zeros_like(d_ctx)
    cb = np.zeros(n, T.COEFF_BITS_JOB_DT)
    for i, c in enumerate(cases):
        cb[i] = (d_lev.data_ptr() + i * 32, d_ctx.data_ptr() + i * 160, d_out.data_ptr() + i * 160, 2, c["ttype"], c["intra"], c["dir"], c["signhide"], 0)
    d_cb = torch.from_numpy(cb.view(np.uint8).copy()).cuda()
    d_bits = torch.zeros(n, dtype=torch.int64, device="cuda")
    assert hip.lib.x265amd_coeff_bits_wave(None, C.c_void_p(d_cb.data_ptr()), n, C.c_void_p(d_bits.data_ptr())) == 0
    torch.cuda.synchronize()
    bits = d_bits.cpu().numpy().astype(np.uint64); out = d_out.cpu().numpy()
    for i in range(n):
        assert int(bits[i]) == want[i][0], (i, cases[i], levels[i], int(bits[i]), want[i][0])
        assert np.array_equal(out[i, :T.CTX_COUNT], want[i][1]), (i, cases[i], levels[i])


def levels_nxn(rng, N, kind):
    lv = np.zeros(N * N, np.int64)
    if kind == 0:           # a few low-frequency levels
        k = int(rng.integers(1, 8))
        ys, xs = rng.integers(0, min(N, 6), k), rng.integers(0, min(N, 6), k)
        lv[ys * N + xs] = rng.integers(1, 5, k) * rng.choice([-1, 1], k)
    elif kind == 1:         # sparse anywhere
        k = int(rng.integers(1, max(2, N * N // 16)))
        idx = rng.choice(N * N, k, replace=False)
        lv[idx] = rng.integers(1, 3, k) * rng.choice([-1, 1], k)
    elif kind == 2:         # dense, small
        m = rng.random(N * N) < 0.6
        lv[m] = (rng.integers(1, 4, int(m.sum())) * rng.choice([-1, 1], int(m.sum())))
    elif kind == 3:         # dense with escapes
        m = rng.random(N * N) < 0.5
        lv[m] = (np.where(rng.random(int(m.sum())) < 0.6, 1, rng.integers(1, 900, int(m.sum()))) * rng.choice([-1, 1], int(m.sum())))
    elif kind == 4:         # one level somewhere (lone last position, groups with an implied flag)
        lv[int(rng.integers(0, N * N))] = int(rng.integers(1, 40)) * int(rng.choice([-1, 1]))
    else:                   # whole groups set / empty
        g = N // 4
        for gy in range(g):
            for gx in range(g):
                r = rng.random()
                if r < 0.3:
                    blk = rng.integers(-3, 4, (4, 4))
                elif r < 0.45:
                    blk = np.zeros((4, 4), np.int64); blk[0, 0] = int(rng.integers(1, 5))
                else:
                    continue
                for y in range(4):
                    lv[(gy * 4 + y) * N + gx * 4:(gy * 4 + y) * N + gx * 4 + 4] = blk[y]
    return lv.astype(np.int16)


@pytest.mark.parametrize("depth", [8, 10])
def test_wave_coeff_bits_all_sizes(depth):
    import torch
    hip, orc = T.load_hip(depth), T.load_oracle(depth)
    rng = np.random.default_rng(4500 + depth)
    cases, levels = [], []
    for log2, count in ((3, 1500), (4, 500), (5, 200)):
        N = 1 << log2
        for i in range(count):
            ttype = int(rng.integers(0, 3)) if log2 < 5 else 0
            c = dict(log2=log2, ttype=ttype, intra=int(rng.random() < 0.7), dir=int(rng.integers(0, 35)), signhide=int(rng.integers(0, 2)),
                     ctx=T.entropy_reset(orc, int(rng.integers(0, 3)), int(rng.integers(10, 46))) if i % 3 else rng.integers(0, 126, T.CTX_COUNT).astype(np.uint8))
            lv = levels_nxn(rng, N, i % 6)
            if not lv.any():
                lv[0] = 1
            cases.append(c); levels.append((int(np.count_nonzero(lv)), lv))
    n = len(cases)
    want = T.coeff_bits_run(orc, cases, levels)
    offs = np.cumsum([0] + [len(l[1]) for l in levels])
    lev = np.concatenate([l[1] for l in levels])
    ctxs = np.zeros((n, 160), np.uint8)
    for i, c in enumerate(cases):
        ctxs[i, :T.CTX_COUNT] = c["ctx"]
    d_lev = torch.from_numpy(lev).cuda(); d_ctx = torch.from_numpy(ctxs).cuda(); d_out = torch.zeros_like(d_ctx)
    cb = np.zeros(n, T.COEFF_BITS_JOB_DT)
    for i, c in enumerate(cases):
        cb[i] = (d_lev.data_ptr() + int(offs[i]) * 2, d_ctx.data_ptr() + i * 160, d_out.data_ptr() + i * 160, c["log2"], c["ttype"], c["intra"], c["dir"], c["signhide"], 0)
    d_cb = torch.from_numpy(cb.view(np.uint8).copy()).cuda()
    d_bits = torch.zeros(n, dtype=torch.int64, device="cuda")
    assert hip.lib.x265amd_coeff_bits_wave(None, C.c_void_p(d_cb.data_ptr()), n, C.c_void_p(d_bits.data_ptr())) == 0
    torch.cuda.synchronize()
    bits = d_bits.cpu().numpy().astype(np.uint64); out = d_out.cpu().numpy()
    for i in range(n):
        assert int(bits[i]) == want[i][0], (i, cases[i]["log2"], cases[i]["ttype"], cases[i]["intra"], cases[i]["dir"], i % 6, int(bits[i]), want[i][0])
        assert np.array_equal(out[i, :T.CTX_COUNT], want[i][1]), (i, cases[i]["log2"], cases[i]["ttype"], i % 6, np.nonzero(out[i, :T.CTX_COUNT] != want[i][1])[0])
