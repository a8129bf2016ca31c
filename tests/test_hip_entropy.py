"""GPU: contexts / estBit / RDOQ / bits-only coefficient coding of libx265amd against the oracle and the golden vectors of the
reference (host-pointer forms), then the batched device entry points (x265amd_est_bit, x265amd_tu_chain_rdoq,
x265amd_coeff_bits) against the oracle on the same cases."""
import ctypes as C

import numpy as np
import pytest

import hevc_testlib as T
from test_entropy_golden import check_entropy


@pytest.mark.parametrize("depth", [8, 10])
def test_product_entropy_reset_host(depth):
    """x265amd_entropy_reset is host arithmetic inside the C-ABI library: checked without a GPU"""
    import os
    gold = np.load(os.path.join(T.GOLDEN_DIR, "entropy_golden.npz"))["reset/%d" % depth]
    L = T.load_hip(depth)
    assert np.array_equal(np.stack([T.entropy_reset(L, st, qp) for st in range(3) for qp in range(52)]), gold)


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [8, 10])
def test_hip_entropy_matches_golden(depth):
    check_entropy(T.load_hip(depth), depth, seeds=(0,))


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [8, 10])
def test_hip_est_bit_batch(depth):
    import torch
    hip, orc = T.load_hip(depth), T.load_oracle(depth)
    rng = np.random.default_rng(9)
    shapes = [(l, 1) for l in range(2, 6)] + [(l, 0) for l in range(2, 5)]
    n = 300
    ctxs = np.zeros((n, 160), np.uint8)
    ctxs[:, :T.CTX_COUNT] = rng.integers(0, 126, (n, T.CTX_COUNT))
    jobs = np.zeros(n, T.EST_JOB_DT)
    d_ctx = torch.from_numpy(ctxs).cuda()
    d_est = torch.zeros(n * T.EST_INTS, dtype=torch.int32, device="cuda")
    want = []
    for i in range(n):
        l, lu = shapes[i % len(shapes)]
        jobs[i] = (d_ctx.data_ptr() + i * 160, d_est.data_ptr() + i * T.EST_INTS * 4, l, lu, 0)
        want.append(T.est_bit(orc, ctxs[i], l, lu))
    d_jobs = torch.from_numpy(jobs.view(np.uint8).copy()).cuda()
    assert hip.lib.x265amd_est_bit(None, C.c_void_p(d_jobs.data_ptr()), n) == 0
    torch.cuda.synchronize()
    assert np.array_equal(d_est.cpu().numpy().reshape(n, T.EST_INTS), np.stack(want))


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [8, 10])
def test_hip_tu_chain_rdoq_and_coeff_bits(depth):
    import torch
    hip, orc = T.load_hip(depth), T.load_oracle(depth)
    dt = np.uint8 if depth == 8 else np.uint16
    isz = np.dtype(dt).itemsize
    f = hip.lib.x265amd_rdoq_lambda
    for seed in range(3):
        cases = T.rdoq_cases(depth, 800 + seed, 300)
        if seed == 2:
            for c in cases[::3]:
                c["rdoq"] = 0          # jobs of one batch may mix RDOQ and plain quantisation
        want = [T.rdoq_chain_oracle(orc, [c])[0] if c["rdoq"] else T.tu_run_chain_oracle(orc, [c])[0] for c in cases]
        n = len(cases)
        per = 32 * 32 * (isz * 3 + 2 * 2)
        arena = np.zeros(n * per, np.uint8)
        ests = np.stack([T.est_bit(orc, c["ctx"], c["log2"], int(c["ttype"] == 0)) for c in cases])
        ctxs = np.zeros((n, 160), np.uint8)
        for i, c in enumerate(cases):
            N = 1 << c["log2"]
            base = i * per
            arena[base:base + 1024 * isz].view(dt).reshape(32, 32)[:N, :N] = c["fenc"]
            arena[base + 1024 * isz:base + 2048 * isz].view(dt).reshape(32, 32)[:N, :N] = c["pred"]
            ctxs[i, :T.CTX_COUNT] = c["ctx"]
        d_arena = torch.from_numpy(arena).cuda()
        d_est = torch.from_numpy(ests).cuda()
        d_ctx = torch.from_numpy(ctxs).cuda()
        d_ctx_out = torch.zeros_like(d_ctx)
        a0 = d_arena.data_ptr()
        jobs = np.zeros(n, T.TU_JOB_DT); rq = np.zeros(n, T.TU_RDOQ_DT); cb = np.zeros(n, T.COEFF_BITS_JOB_DT)
        for i, c in enumerate(cases):
            base = a0 + i * per
            jobs[i] = (base, base + 1024 * isz, base + 3072 * isz, base + 3072 * isz + 2048, base + 2048 * isz, 32, 32, 32, 32,
                       c["log2"], c["ttype"], c["intra"], c["dir"], c["slice"], c["qp"], c["signhide"], 0)
            l2, l1 = C.c_int64(0), C.c_int32(0)
            f(c["qp"], C.byref(l2), C.byref(l1))
            rq[i] = (d_est.data_ptr() + i * T.EST_INTS * 4, l2.value, l1.value, c["psyrdoq"], c["rdoq"], c["tudepth"], 0)
            cb[i] = (base + 3072 * isz, d_ctx.data_ptr() + i * 160, d_ctx_out.data_ptr() + i * 160, c["log2"], c["ttype"], c["intra"], c["dir"], c["signhide"], 0)
        d_jobs = torch.from_numpy(jobs.view(np.uint8).copy()).cuda()
        d_rq = torch.from_numpy(rq.view(np.uint8).copy()).cuda()
        d_cb = torch.from_numpy(cb.view(np.uint8).copy()).cuda()
        d_out = torch.zeros(n * T.TU_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda")
        d_bits = torch.zeros(n, dtype=torch.int64, device="cuda")
        assert hip.lib.x265amd_tu_chain_rdoq(None, C.c_void_p(d_jobs.data_ptr()), C.c_void_p(d_rq.data_ptr()), n, C.c_void_p(d_out.data_ptr())) == 0
        assert hip.lib.x265amd_coeff_bits(None, C.c_void_p(d_cb.data_ptr()), n, C.c_void_p(d_bits.data_ptr())) == 0
        torch.cuda.synchronize()
        res = d_out.cpu().numpy().view(T.TU_RESULT_DT)
        back = d_arena.cpu().numpy()
        bits = d_bits.cpu().numpy().astype(np.uint64)
        ctx_out = d_ctx_out.cpu().numpy()
        wbits = T.coeff_bits_run(orc, cases, [(int(w[0][0]), w[1]) for w in want])
        for i, (c, w) in enumerate(zip(cases, want)):
            N = 1 << c["log2"]
            st, coeff, resi, recon = w
            got = (int(res[i]["num_sig"]), int(res[i]["zero_dist"]), int(res[i]["zero_energy"]), int(res[i]["nz_dist"]), int(res[i]["nz_energy"]))
            assert got == tuple(int(v) for v in st), (i, got, st)
            base = i * per
            assert np.array_equal(back[base + 3072 * isz:base + 3072 * isz + N * N * 2].view(np.int16), coeff), i
            assert np.array_equal(back[base + 2048 * isz:base + 3072 * isz].view(dt).reshape(32, 32)[:N, :N], recon), i
            assert np.array_equal(back[base + 3072 * isz + 2048:base + 3072 * isz + 4096].view(np.int16).reshape(32, 32)[:N, :N], resi), i
            assert int(bits[i]) == wbits[i][0], (i, int(bits[i]), wbits[i][0])
            assert np.array_equal(ctx_out[i, :T.CTX_COUNT], wbits[i][1]), i
