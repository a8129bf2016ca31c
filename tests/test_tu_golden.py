"""Residual path against golden results of the reference's Quant / RDCost classes (tests/golden/tu_golden.npz):
the oracle (always), and the product's host-side RDCost formulas (pure host code, no GPU needed)."""
import ctypes as C
import os

import numpy as np
import pytest

import hevc_testlib as T

GOLD = np.load(os.path.join(T.GOLDEN_DIR, "tu_golden.npz"))


def check_tu(L, depth):
    for seed in range(4):
        cases = T.tu_cases(depth, 100 + seed, 150)
        res = T.tu_run_host(L, cases)
        assert np.array_equal(np.array([r[0] for r in res], np.int32), GOLD["tu/%d/%d/numsig" % (depth, seed)])
        assert np.array_equal(np.concatenate([r[1] for r in res]), GOLD["tu/%d/%d/coeff" % (depth, seed)])
        assert np.array_equal(np.concatenate([r[2].ravel() for r in res]), GOLD["tu/%d/%d/resi" % (depth, seed)])


def check_rdcost(lib, fn, depth):
    rng = np.random.default_rng(5)
    want = GOLD["rdcost/%d" % depth]
    for k in range(300):
        qp, st = int(rng.integers(0, 70)), int(rng.integers(0, 3))
        psy = float(rng.choice([0.0, 1.0, 2.0, 0.7]))
        dist, bits, pc = int(rng.integers(0, 1 << 24)), int(rng.integers(0, 1 << 16)), int(rng.integers(0, 1 << 16))
        a = np.zeros(6, np.uint64)
        getattr(lib, fn)(qp, st, C.c_double(psy), C.c_uint64(dist), C.c_uint32(bits), C.c_uint32(pc), T._ptr(a))
        assert np.array_equal(a, want[k]), (k, qp, st, psy)


@pytest.mark.parametrize("depth", [8, 10])
def test_oracle_tu(depth):
    check_tu(T.load_oracle(depth), depth)


@pytest.mark.parametrize("depth", [8, 10])
def test_oracle_rdcost(depth):
    check_rdcost(T.load_oracle(depth).lib, "orc_rdcost", depth)


@pytest.mark.parametrize("depth", [8, 10])
def test_product_rdcost_host(depth):
    """x265amd_rdcost is host arithmetic inside the C-ABI library: checked without a GPU"""
    check_rdcost(T.load_hip(depth).lib, "x265amd_rdcost", depth)


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [8, 10])
def test_hip_transform_inverse(depth):
    check_tu(T.load_hip(depth), depth)


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [8, 10])
def test_hip_tu_chain(depth):
    """the fused per-TU kernel (x265amd_tu_chain) on a batch, against the oracle's restatement of the measurement"""
    import torch
    hip, orc = T.load_hip(depth), T.load_oracle(depth)
    dt = np.uint8 if depth == 8 else np.uint16
    for seed in range(3):
        cases = T.tu_cases(depth, 300 + seed, 200)
        want = T.tu_run_chain_oracle(orc, cases)
        # one arena: per case fenc | pred | coeff | resi | recon at stride 32
        isz = np.dtype(dt).itemsize
        per = 32 * 32 * (isz * 3 + 2 * 2)
        arena = np.zeros(len(cases) * per, np.uint8)
        jobs = np.zeros(len(cases), T.TU_JOB_DT)
        for i, c in enumerate(cases):
            N = 1 << c["log2"]
            base = i * per
            f = arena[base:base + 1024 * isz].view(dt).reshape(32, 32); f[:N, :N] = c["fenc"]
            p = arena[base + 1024 * isz:base + 2048 * isz].view(dt).reshape(32, 32); p[:N, :N] = c["pred"]
            jobs[i] = (0, 0, 0, 0, 0, 32, 32, 32, 32, c["log2"], c["ttype"], c["intra"], c["dir"], c["slice"], c["qp"], c["signhide"], 0)
        d_arena = torch.from_numpy(arena).cuda()
        a0 = d_arena.data_ptr()
        for i in range(len(cases)):
            base = a0 + i * per
            jobs[i]["fenc"] = base; jobs[i]["pred"] = base + 1024 * isz
            jobs[i]["recon"] = base + 2048 * isz; jobs[i]["coeff"] = base + 3072 * isz; jobs[i]["resi"] = base + 3072 * isz + 2048
        d_jobs = torch.from_numpy(jobs.view(np.uint8).copy()).cuda()
        d_out = torch.zeros(len(cases) * T.TU_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda")
        rc = hip.lib.x265amd_tu_chain(None, C.c_void_p(d_jobs.data_ptr()), len(cases), C.c_void_p(d_out.data_ptr()))
        assert rc == 0
        torch.cuda.synchronize()
        res = d_out.cpu().numpy().view(T.TU_RESULT_DT)
        back = d_arena.cpu().numpy()
        for i, (c, w) in enumerate(zip(cases, want)):
            N = 1 << c["log2"]
            st, coeff, resi, recon = w
            got = (int(res[i]["num_sig"]), int(res[i]["zero_dist"]), int(res[i]["zero_energy"]), int(res[i]["nz_dist"]), int(res[i]["nz_energy"]))
            assert got == tuple(int(v) for v in st), (i, got, st)
            base = i * per
            assert np.array_equal(back[base + 2048 * isz:base + 3072 * isz].view(dt).reshape(32, 32)[:N, :N], recon), i
            assert np.array_equal(back[base + 3072 * isz:base + 3072 * isz + N * N * 2].view(np.int16), coeff), i
            assert np.array_equal(back[base + 3072 * isz + 2048:base + 3072 * isz + 4096].view(np.int16).reshape(32, 32)[:N, :N], resi), i
