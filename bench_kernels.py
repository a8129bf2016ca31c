#!/usr/bin/env python3
"""bench.py -- hot-path throughput of the MI355X HEVC encode path on BASELINE.json's metric/config.

A "step" is one pass of the hot path over one synthetic 1920x1080 8-bit frame (configs[1], --preset medium search
parameters: hex search, merange 57, subme 2, 3 reference pictures): motion estimation of every 2Nx2N PU of every
CTU (64x64 ... 8x8) against every reference through the fused kernel `x265amd_me_search`, exactly as the reference's
MotionEstimate::motionEstimate() would be called for those PUs.  Inputs are resident in HBM before the timed region.
This measures the hot-path kernels, NOT a full encode: entropy coding, mode decision and the other rows of
SURVEY.md section 8 are not in the timed region yet (DESIGN.md "what the bench measures").

Contract: python bench.py --gpus N --steps K --warmup W ; prints ONE JSON line on rank 0.
Multi-GPU: one process per GPU (torch.distributed, backend nccl == RCCL); frames are sharded one per GPU and each
finished frame is all-gathered so that every rank holds it as a future reference picture (weak scaling).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H, DEPTH = 1920, 1080, 8
MARGIN_X, MARGIN_Y = 96, 80          # PicYuv margins for CTU 64 (reference: common/picyuv.cpp create)
NUM_REFS, MERANGE, SUBME, QP = 3, 57, 2, 32
MAX_WIN = (192, 192)
HBM_PEAK_GBS = 8000.0                # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec



class _PlaneExchange:
    """Whole-picture exchange between GPUs for THIS kernel workload only (--gpus N): picture k in encode order belongs to rank k % world, and after a step every rank holds
    the padded plane set of every rank's picture (one all_gather per step over torch.distributed).  The encoder does not work this way: its ranks publish finished CTU ROWS
    (x265-amod_amd/frame_rows.py, SURVEY.md section 8e as written)."""

    @staticmethod
    def frames_of_step(step, world):
        return [step * world + r for r in range(world)]

    class ReferenceRing:
        def __init__(self, depth):
            self.depth = depth
            self.pics = {}

        def put(self, idx, plane):
            self.pics[idx] = plane
            for k in sorted(self.pics):
                if len(self.pics) <= self.depth:
                    break
                del self.pics[k]

        def get(self, idx):
            return self.pics[idx]

        def has(self, idx):
            return idx in self.pics

    @staticmethod
    def publish_step(local_plane, step, ring, gather_buf=None):
        import torch
        import torch.distributed as dist
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        if world == 1:
            ring.put(step, local_plane)
            return None
        if gather_buf is None:
            gather_buf = torch.empty(world * local_plane.numel(), dtype=local_plane.dtype, device=local_plane.device)
        dist.all_gather_into_tensor(gather_buf, local_plane.reshape(-1))
        n = local_plane.numel()
        for r, idx in enumerate(_PlaneExchange.frames_of_step(step, world)):
            ring.put(idx, gather_buf[r * n:(r + 1) * n].clone() if r != rank else local_plane)
        return gather_buf


def lcg_noise(shape, seed):
    """integer-only noise in [-12, 12] (SURVEY.md section 8d generator)"""
    n = shape[0] * shape[1]
    idx = np.arange(n, dtype=np.uint64)
    v = idx * np.uint64(6364136223846793005) + np.uint64((int(seed) * 1442695040888963407) & 0xFFFFFFFFFFFFFFFF)         # (array arithmetic wraps modulo 2^64; the scalar product is reduced here)
    v ^= v >> np.uint64(33)
    v = (v * np.uint64(0xFF51AFD7ED558CCD)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    v ^= v >> np.uint64(29)
    return ((v % np.uint64(25)).astype(np.int64) - 12).reshape(shape)


def make_clip(nframes):
    """padded planes (Y, U, V) of a moving textured gradient: frame t is the base shifted by (2t, t) luma samples"""
    bw, bh = W + 2 * nframes + 8, H + nframes + 8
    yy, xx = np.mgrid[0:bh, 0:bw].astype(np.int64)
    base = 128 + ((xx * 3 + yy * 2) % 160 - 80) // 2 + (((xx >> 4) ^ (yy >> 4)) & 7) * 6 + lcg_noise((bh, bw), 0x9E3779B9 ^ (2 << 8))
    base = np.clip(base, 0, 255)
    cy, cx = np.mgrid[0:bh // 2, 0:bw // 2].astype(np.int64)
    baseU = np.clip(96 + (cx + cy) % 64 + lcg_noise((bh // 2, bw // 2), 77) // 3, 0, 255)
    baseV = np.clip(160 - (cx * 2 + cy) % 48 + lcg_noise((bh // 2, bw // 2), 78) // 3, 0, 255)
    frames = []
    for t in range(nframes):
        pic = np.clip(base[t:t + H, 2 * t:2 * t + W] + lcg_noise((H, W), 1000 + t) // 4, 0, 255).astype(np.uint8)
        u = np.clip(baseU[t // 2:t // 2 + H // 2, t:t + W // 2] + lcg_noise((H // 2, W // 2), 2000 + t) // 6, 0, 255).astype(np.uint8)
        v = np.clip(baseV[t // 2:t // 2 + H // 2, t:t + W // 2] + lcg_noise((H // 2, W // 2), 3000 + t) // 6, 0, 255).astype(np.uint8)
        frames.append((np.pad(pic, ((MARGIN_Y, MARGIN_Y), (MARGIN_X, MARGIN_X)), mode="edge"),
                       np.pad(u, ((MARGIN_Y // 2, MARGIN_Y // 2), (MARGIN_X // 2, MARGIN_X // 2)), mode="edge"),
                       np.pad(v, ((MARGIN_Y // 2, MARGIN_Y // 2), (MARGIN_X // 2, MARGIN_X // 2)), mode="edge")))
    return frames


def cu_grid(size):
    """top-left corners of every size x size CU that lies inside the picture"""
    ys, xs = np.mgrid[0:H - size + 1:size, 0:W - size + 1:size]
    return xs.ravel().astype(np.int64), ys.ravel().astype(np.int64)


def tu_jobs(T, cur, ref, arena_base):
    """residual measurements of one frame: for every CU of size 32/16/8 the luma TU and both chroma TUs, prediction = the
    previous picture displaced by the true motion (inter P-slice, sign hiding on).  Returns (jobs, arena_bytes)."""
    stride, stride_c = W + 2 * MARGIN_X, W // 2 + MARGIN_X
    org, org_c = MARGIN_Y * stride + MARGIN_X, (MARGIN_Y // 2) * stride_c + MARGIN_X // 2
    parts, off = [], 0
    for size in (32, 16, 8):
        x, y = cu_grid(size)
        for plane, (n, st, o, mvx, mvy, qp) in enumerate(((size, stride, org, 2, 1, QP), (size // 2, stride_c, org_c, 1, 0, 31), (size // 2, stride_c, org_c, 1, 0, 31))):
            sx, sy = (x, y) if plane == 0 else (x // 2, y // 2)
            j = np.zeros(len(x), T.TU_JOB_DT)
            j["fenc"] = cur[plane] + o + sy * st + sx
            j["pred"] = ref[plane] + o + (sy + mvy) * st + sx + mvx
            per = n * n * 5                                   # recon n*n, coeff 2*n*n, resi 2*n*n
            base = arena_base + off + np.arange(len(x), dtype=np.int64) * per
            j["recon"], j["coeff"], j["resi"] = base, base + n * n, base + 3 * n * n
            j["fenc_stride"], j["pred_stride"], j["resi_stride"], j["recon_stride"] = st, st, n, n
            j["log2"], j["ttype"], j["intra"], j["dir"], j["slice"], j["qp"], j["signhide"] = int(np.log2(n)), plane, 0, 0, 1, qp, 1
            off += len(x) * per
            parts.append(j)
    return np.concatenate(parts), off


def intra_jobs(T, cur, ref):
    """35-mode scans of one frame: every CU of size 32/16/8; the previous picture stands in for the reconstruction"""
    stride = W + 2 * MARGIN_X
    org = MARGIN_Y * stride + MARGIN_X
    parts = []
    for size in (32, 16, 8):
        x, y = cu_grid(size)
        u = size // 4
        left, above = x > 0, y > 0
        above_right = above & (x + 2 * size <= W) & (((x // size) & 1) == 0)
        mask = np.zeros(len(x), np.uint64)
        ones = lambda k: np.uint64((1 << k) - 1)
        mask |= np.where(left, ones(u) << np.uint64(u), np.uint64(0))
        mask |= np.where(left & above, np.uint64(1) << np.uint64(2 * u), np.uint64(0))
        mask |= np.where(above, ones(u) << np.uint64(2 * u + 1), np.uint64(0))
        mask |= np.where(above_right, ones(u) << np.uint64(3 * u + 1), np.uint64(0))
        j = np.zeros(len(x), T.INTRA_JOB_DT)
        j["recon"] = ref[0] + org + y * stride + x
        j["fenc"] = cur[0] + org + y * stride + x
        j["avail"] = mask
        j["recon_stride"], j["fenc_stride"], j["log2"], j["strong"] = stride, stride, int(np.log2(size)), 1
        parts.append(j)
    return np.concatenate(parts)


def inter_cost_jobs(T, cur_idx, arena_base):
    """merge-candidate scan of one frame (checkMerge2Nx2N_rd0_4): for every 2Nx2N CU 64..8 two candidates predicted from the
    previous picture (uni-prediction, P slice) at the true motion and one quarter sample off it, measured with SA8D
    (+ chroma SA8D for CUs >= 16).  Returns (jobs, arena_bytes)."""
    parts, off = [], 0
    for size in (64, 32, 16, 8):
        x, y = cu_grid(size)
        for cand in range(2):
            j = np.zeros(len(x), T.MC_JOB_DT)
            chroma = size >= 16
            per = size * size + (size * size // 2 if chroma else 0)
            base = arena_base + off + np.arange(len(x), dtype=np.int64) * per
            j["dstY"], j["dstU"], j["dstV"] = base, base + size * size, base + size * size + size * size // 4
            j["dstStride"], j["dstCStride"] = size, size // 2
            j["x"], j["y"], j["cuX"], j["cuY"], j["w"], j["h"] = x, y, x, y, size, size
            j["ref0"], j["ref1"] = cur_idx - 1, -1
            j["mv0"][:, 0], j["mv0"][:, 1] = -8 + cand, -4 - cand
            j["sliceType"], j["flags"], j["metric"], j["chroma_cost"] = 1, 3 if chroma else 1, 3, 1 if chroma else 0
            off += len(x) * per
            parts.append(j)
    return np.concatenate(parts), off


def intra_tu_jobs(T, cur, ref, arena_base):
    """intra TU coding step of one frame: for every CU of size 32/16/8 one luma TU (mode varies with the position) and both
    chroma TUs (DC / planar), neighbours from the previous picture standing in for the reconstruction"""
    stride, stride_c = W + 2 * MARGIN_X, W // 2 + MARGIN_X
    org, org_c = MARGIN_Y * stride + MARGIN_X, (MARGIN_Y // 2) * stride_c + MARGIN_X // 2
    parts, off = [], 0
    for size in (32, 16, 8):
        x, y = cu_grid(size)
        left, above = x > 0, y > 0
        above_right = above & (x + 2 * size <= W) & (((x // size) & 1) == 0)
        for plane, (n, st, o, qp) in enumerate(((size, stride, org, QP), (size // 2, stride_c, org_c, 31), (size // 2, stride_c, org_c, 31))):
            u = n // 4
            ones = lambda k: np.uint64((1 << k) - 1)
            mask = np.zeros(len(x), np.uint64)
            mask |= np.where(left, ones(u) << np.uint64(u), np.uint64(0))
            mask |= np.where(left & above, np.uint64(1) << np.uint64(2 * u), np.uint64(0))
            mask |= np.where(above, ones(u) << np.uint64(2 * u + 1), np.uint64(0))
            mask |= np.where(above_right, ones(u) << np.uint64(3 * u + 1), np.uint64(0))
            sx, sy = (x, y) if plane == 0 else (x // 2, y // 2)
            j = np.zeros(len(x), T.INTRA_TU_JOB_DT)
            t = j["tu"]
            per = n * n * 5
            base = arena_base + off + np.arange(len(x), dtype=np.int64) * per
            t["fenc"] = cur[plane] + o + sy * st + sx
            t["pred"] = 0
            t["recon"], t["coeff"], t["resi"] = base, base + n * n, base + 3 * n * n
            t["fenc_stride"], t["pred_stride"], t["resi_stride"], t["recon_stride"] = st, n, n, n
            t["log2"], t["ttype"], t["intra"], t["slice"], t["qp"], t["signhide"] = int(np.log2(n)), plane, 1, 1, qp, 1
            t["dir"] = ((x // size) * 7 + (y // size) * 3) % 35 if plane == 0 else 1 - (plane & 1)
            j["tu"] = t
            j["nb"] = ref[plane] + o + sy * st + sx
            j["avail"], j["nb_stride"], j["strong"] = mask, st, 1
            off += len(x) * per
            parts.append(j)
    return np.concatenate(parts), off


def frame_jobs(T, refdist):
    """every 2Nx2N PU of every CTU that lies inside the picture, for one reference at temporal distance refdist"""
    jobs = []
    rng = np.random.default_rng(1234 + refdist)
    tmx, tmy = -2 * refdist, -refdist            # true motion towards the older frame (full-pel)
    for cy in range(0, H, 64):
        for cx in range(0, W, 64):
            for size in (64, 32, 16, 8):
                for y in range(cy, min(cy + 64, H), size):
                    for x in range(cx, min(cx + 64, W), size):
                        if x + size > W or y + size > H:
                            continue
                        mvp = (tmx * 4 + int(rng.integers(-5, 6)), tmy * 4 + int(rng.integers(-5, 6)))
                        mnx = max((mvp[0] >> 2) - MERANGE, -x - 64); mxx = min((mvp[0] >> 2) + MERANGE, W - x - size + 64)
                        mny = max((mvp[1] >> 2) - MERANGE, -y - 56); mxy = min((mvp[1] >> 2) + MERANGE, H - y - size + 56)
                        mvc = [(int(rng.integers(-16, 17)), int(rng.integers(-16, 17))), (tmx * 4, tmy * 4)]
                        jobs.append(dict(x=x, y=y, w=size, h=size, qp=QP, mvp=mvp, mvmin=(mnx, mny), mvmax=(mxx, mxy), mvc=mvc,
                                         merange=MERANGE, method=T.ME_HEX, subme=SUBME))
    return jobs


def cpu_baseline(T, frames, packed_by_ref, budget_s=20.0):
    """the SAME frame workload on ONE host core, timed inside C loops: through the reference's own MotionEstimate / Predict /
    Quant classes and primitives (oracle/_ref, kind "reference") when that build is present, else through the oracle port.
    Bounded: each of the three parts is cut off at its share of the budget and extrapolated."""
    if T.have_ref():
        L, kind = T.load_ref(DEPTH), "reference"
    else:
        L, kind = T.load_oracle(DEPTH), "port"
    stride = W + 2 * MARGIN_X
    origin = MARGIN_Y * stride + MARGIN_X
    cur, prev = frames[NUM_REFS], frames[NUM_REFS - 1]
    hcur = np.concatenate([p.ravel() for p in cur]); hprev = np.concatenate([p.ravel() for p in prev])
    ysz, csz = cur[0].size, cur[1].size
    addr = lambda h: (h.ctypes.data, h.ctypes.data + ysz, h.ctypes.data + ysz + csz)
    secs_per_frame, notes = 0.0, []
    # motion searches
    total, done, spent = sum(len(j) for j in packed_by_ref), 0, 0.0
    for r, pk in enumerate(packed_by_ref):
        ref = frames[NUM_REFS - 1 - r][0].ravel()
        for k in range(0, len(pk), 8192):
            t0 = time.perf_counter()
            T.me_run_host_batch(L, cur[0].ravel(), ref, stride, origin, pk[k:k + 8192])
            spent += time.perf_counter() - t0; done += len(pk[k:k + 8192])
            if spent > budget_s / 3 * (r + 1) / NUM_REFS:
                break
    secs_per_frame += spent * total / done; notes.append("%d/%d searches %.1fs" % (done, total, spent))
    # intra scans
    ij = intra_jobs(T, addr(hcur), addr(hprev))
    out = np.zeros(len(ij) * 35, np.int32)
    fn = getattr(L.lib, L.prefix + "intra_scan_batch")
    done, spent = 0, 0.0
    for k in range(0, len(ij), 2048):
        t0 = time.perf_counter()
        fn(T._ptr(np.ascontiguousarray(ij[k:k + 2048])), len(ij[k:k + 2048]), T.off(out, 35 * k))
        spent += time.perf_counter() - t0; done += len(ij[k:k + 2048])
        if spent > budget_s / 3:
            break
    secs_per_frame += spent * len(ij) / done; notes.append("%d/%d intra scans %.1fs" % (done, len(ij), spent))
    # TU chains
    arena = np.zeros(1, np.uint8)
    tj, nbytes = tu_jobs(T, addr(hcur), addr(hprev), 0)
    arena = np.zeros(nbytes, np.uint8)
    for f in ("recon", "coeff", "resi"):
        tj[f] += arena.ctypes.data
    tout = np.zeros(len(tj), T.TU_RESULT_DT)
    fn = getattr(L.lib, L.prefix + "tu_chain_batch")
    done, spent = 0, 0.0
    order = np.random.default_rng(0).permutation(len(tj))       # sizes are grouped in the list: sample them evenly
    tj = np.ascontiguousarray(tj[order])
    for k in range(0, len(tj), 4096):
        t0 = time.perf_counter()
        fn(T._ptr(tj[k:k + 4096]), len(tj[k:k + 4096]), T.off(tout.view(np.uint8), 32 * k))
        spent += time.perf_counter() - t0; done += len(tj[k:k + 4096])
        if spent > budget_s / 3:
            break
    secs_per_frame += spent * len(tj) / done; notes.append("%d/%d TU chains %.1fs" % (done, len(tj), spent))
    # coefficient bits of the TU chains that were run
    ndone = done
    ctx0 = np.zeros(160, np.uint8)
    getattr(L.lib, L.prefix + "entropy_reset")(1, QP, T._ptr(ctx0))
    ctx_out = np.zeros((ndone, 160), np.uint8)
    cb = np.zeros(ndone, T.COEFF_BITS_JOB_DT)
    cb["coeff"], cb["ctx_in"], cb["ctx_out"] = tj["coeff"][:ndone], ctx0.ctypes.data, ctx_out.ctypes.data + np.arange(ndone, dtype=np.int64) * 160
    cb["log2"], cb["ttype"], cb["signhide"] = tj["log2"][:ndone], tj["ttype"][:ndone], 1
    bits = np.zeros(ndone, np.uint64)
    fn = getattr(L.lib, L.prefix + "coeff_bits_batch")
    t0 = time.perf_counter()
    fn(T._ptr(cb), ndone, T._ptr(bits))
    spent = time.perf_counter() - t0
    secs_per_frame += spent * len(tj) / ndone; notes.append("%d/%d coefficient codings %.1fs" % (ndone, len(tj), spent))
    # merge-candidate costs
    stride_c = W // 2 + MARGIN_X
    origin_c = (MARGIN_Y // 2) * stride_c + MARGIN_X // 2
    ic, nbytes = inter_cost_jobs(T, 1, 0)
    icarena = np.zeros(nbytes, np.uint8)
    for f in ("dstY", "dstU", "dstV"):
        ic[f] += icarena.ctypes.data
    ic = np.ascontiguousarray(ic[np.random.default_rng(1).permutation(len(ic))])
    pl = np.array([hprev.ctypes.data + origin, hprev.ctypes.data + ysz + origin_c, hprev.ctypes.data + ysz + csz + origin_c], np.uint64)
    fp = np.array([hcur.ctypes.data + origin, hcur.ctypes.data + ysz + origin_c, hcur.ctypes.data + ysz + csz + origin_c], np.uint64)
    cost = np.zeros((len(ic), 2), np.uint32)
    fn = getattr(L.lib, L.prefix + "inter_cost_batch")
    done, spent = 0, 0.0
    for k in range(0, len(ic), 4096):
        t0 = time.perf_counter()
        fn(T._ptr(pl), C.c_int64(stride), C.c_int64(stride_c), W, H, T._ptr(ic[k:k + 4096]), len(ic[k:k + 4096]), T._ptr(fp), C.c_int64(stride), C.c_int64(stride_c),
           T.off(cost.view(np.uint8).ravel(), 8 * k))
        spent += time.perf_counter() - t0; done += len(ic[k:k + 4096])
        if spent > budget_s / 4:
            break
    secs_per_frame += spent * len(ic) / done; notes.append("%d/%d merge-candidate costs %.1fs" % (done, len(ic), spent))
    # intra TU steps
    it, nbytes = intra_tu_jobs(T, addr(hcur), addr(hprev), 0)
    itarena = np.zeros(nbytes, np.uint8)
    for f in ("recon", "coeff", "resi"):
        it["tu"][f] += itarena.ctypes.data
    it = np.ascontiguousarray(it[np.random.default_rng(2).permutation(len(it))])
    res = np.zeros(len(it), T.TU_RESULT_DT)
    fn = getattr(L.lib, L.prefix + "intra_tu_chain_batch")
    done, spent = 0, 0.0
    for k in range(0, len(it), 4096):
        t0 = time.perf_counter()
        fn(T._ptr(it[k:k + 4096]), None, len(it[k:k + 4096]), T.off(res.view(np.uint8), 32 * k))
        spent += time.perf_counter() - t0; done += len(it[k:k + 4096])
        if spent > budget_s / 4:
            break
    secs_per_frame += spent * len(it) / done; notes.append("%d/%d intra TU steps %.1fs" % (done, len(it), spent))
    # in-loop filters of one picture (the reference's Deblock / SAO classes on a CUData fixture when kind == "reference")
    hflt = (H // 8) * 8
    dbc = T.deblock_case(DEPTH, 9, W, hflt, True, False)
    sac = T.sao_case(DEPTH, 7, W, hflt)
    t0 = time.perf_counter()
    T.deblock_run_host(L, dbc)
    T.sao_run_host(L, sac)
    spent = time.perf_counter() - t0
    secs_per_frame += spent; notes.append("in-loop filters %.2fs (includes building the CUData fixture)" % spent)
    return {"value": 1.0 / secs_per_frame, "unit": "frames/s", "cores": 1, "kind": kind,
            "sample": "one %dx%d frame of the same workload on one core: " % (W, H) + ", ".join(notes) + "; parts cut at their budget are extrapolated"}


def sample_more(T, orc, frames, last, d_ic, d_ic_out, d_ic_arena, d_it, d_it_out, d_it_arena, d_cb, d_bits, d_ctx_out, d_arena, ctx0, d_pics, planes):
    """oracle checks of samples of the merge-candidate costs, intra TU steps and coefficient bits of the last timed frame: the jobs'
    device addresses are rebased onto host copies of the pictures and arenas and run through the oracle's batch forms"""
    stride, stride_c = W + 2 * MARGIN_X, W // 2 + MARGIN_X
    origin, origin_c = MARGIN_Y * stride + MARGIN_X, (MARGIN_Y // 2) * stride_c + MARGIN_X // 2
    ysz, csz = frames[0][0].size, frames[0][1].size
    host = {k: np.concatenate([f.ravel() for f in frames[k]]) for k in (last, last - 1)}
    dev = {k: d_pics[k].data_ptr() for k in (last, last - 1)}
    ok, checked = True, 0

    def rebase(a, which):
        return a - dev[which] + host[which].ctypes.data

    # merge-candidate costs
    ic = d_ic.cpu().numpy().view(T.MC_JOB_DT)
    sel = np.arange(0, len(ic), 1009)
    jb = ic[sel].copy()
    buf = np.zeros((len(sel), 64 * 64 * 2), np.uint8)
    jb["dstY"] = buf.ctypes.data + np.arange(len(sel)) * buf.shape[1]
    jb["dstU"] = jb["dstY"] + 64 * 64; jb["dstV"] = jb["dstY"] + 64 * 64 + 32 * 32
    jb["ref0"] = 0
    pl = np.array([host[last - 1].ctypes.data + origin, host[last - 1].ctypes.data + ysz + origin_c, host[last - 1].ctypes.data + ysz + csz + origin_c], np.uint64)
    fp = np.array([host[last].ctypes.data + origin, host[last].ctypes.data + ysz + origin_c, host[last].ctypes.data + ysz + csz + origin_c], np.uint64)
    want = np.zeros((len(sel), 2), np.uint32)
    orc.lib.orc_inter_cost_batch(T._ptr(pl), C.c_int64(stride), C.c_int64(stride_c), W, H, T._ptr(jb), len(sel), T._ptr(fp), C.c_int64(stride), C.c_int64(stride_c), T._ptr(want))
    got = d_ic_out.cpu().numpy().view(np.uint32).reshape(-1, 2)[sel]
    ok &= bool(np.array_equal(want, got)); checked += len(sel)
    # intra TU steps
    it = d_it.cpu().numpy().view(T.INTRA_TU_JOB_DT)
    sel = np.arange(0, len(it), 997)
    jb = it[sel].copy()
    t = jb["tu"]
    t["fenc"] = rebase(t["fenc"].astype(np.int64), last).astype(np.uint64)
    out = np.zeros((len(sel), 32 * 32 * 5), np.uint8)
    n2 = (1 << t["log2"].astype(np.int64)) ** 2
    t["recon"] = out.ctypes.data + np.arange(len(sel)) * out.shape[1]
    t["coeff"] = t["recon"] + n2; t["resi"] = t["recon"] + 3 * n2
    jb["tu"] = t
    jb["nb"] = rebase(jb["nb"].astype(np.int64), last - 1).astype(np.uint64)
    res = np.zeros(len(sel), T.TU_RESULT_DT)
    orc.lib.orc_intra_tu_chain_batch(T._ptr(jb), None, len(sel), T._ptr(res))
    gres = d_it_out.cpu().numpy().view(T.TU_RESULT_DT)[sel]
    arena = d_it_arena.cpu().numpy()
    for k, i in enumerate(sel):
        ok &= bool(res[k].tobytes() == gres[k].tobytes())
        o = int(it[i]["tu"]["recon"]) - d_it_arena.data_ptr()
        ok &= bool(np.array_equal(arena[o:o + 5 * int(n2[k])], out[k, :5 * int(n2[k])]))
    checked += len(sel)
    # coefficient bits (on the levels the TU chain kernel has just written)
    cb = d_cb.cpu().numpy().view(T.COEFF_BITS_JOB_DT)
    sel = np.arange(0, len(cb), 991)
    jb = cb[sel].copy()
    tarena = d_arena.cpu().numpy()
    ctx_in = np.ascontiguousarray(ctx0); ctx_out = np.zeros((len(sel), 160), np.uint8)
    jb["coeff"] = jb["coeff"] - np.uint64(d_arena.data_ptr()) + np.uint64(tarena.ctypes.data)
    jb["ctx_in"] = ctx_in.ctypes.data
    jb["ctx_out"] = ctx_out.ctypes.data + np.arange(len(sel)) * 160
    bits = np.zeros(len(sel), np.uint64)
    orc.lib.orc_coeff_bits_batch(T._ptr(jb), len(sel), T._ptr(bits))
    ok &= bool(np.array_equal(bits, d_bits.cpu().numpy().view(np.uint64)[sel]))
    ok &= bool(np.array_equal(ctx_out[:, :T.CTX_COUNT], d_ctx_out.cpu().numpy().reshape(-1, 160)[sel][:, :T.CTX_COUNT]))
    checked += len(sel)
    return ok, checked


def measured_traffic(kernel="k_me_search"):
    """HBM bytes per launch of a kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, run separately:
    counters cannot be read inside the bench); the summary is committed under profiles/"""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_analysis13_traffic.json")) as f:
            return json.load(f)[kernel]["hbm_bytes_per_launch_uncorrected"]
    except (OSError, KeyError, ValueError):
        return None


def encoder_pipeline_sample(T):
    """The real encoder (x265amd_encoder_open / encode: C++ host loop over x265amd_analyse_frame + deblocking + SAO + slice NAL units) on the
    small clip whose output is pinned against the reference ENCODER (tests/golden/frame_pipeline_golden.npz): reported beside the kernel
    workload, never as `value` -- it is bit-exact with the reference's byte stream but analyses one CTU at a time with a synchronous launch
    per block operation (the reference's decision chain is serial), so it measures launch latency."""
    import hashlib
    try:
        g = np.load(os.path.join(ROOT, "tests", "golden", "frame_pipeline_golden.npz"))
        tag = "sao_bframes/"
        L = T.load_hip(8)
        frames, stride, cstride, org = T.frame_clip_b(8)
        planes = [T.frame_planes(f, stride, cstride, org) for f in frames]
        cfg = dict(fpsNum=30, fpsDenom=1, qp=30, aspectRatioIdc=1, bframes=2, bEnableLoopFilter=1, bEnableSAO=1, bEnableWavefront=0)
        T.encoder_run(L, planes, T.MC_W, T.MC_H, **cfg)     # warm-up
        t0 = time.perf_counter()
        stream, coded = T.encoder_run(L, planes, T.MC_W, T.MC_H, **cfg)
        dt = time.perf_counter() - t0
        want = g[tag + "stream"]
        same = len(stream) == len(want) and hashlib.md5(stream.tobytes()).hexdigest() == hashlib.md5(want.tobytes()).hexdigest()
        out = {"clip": "256x192 8-bit, 7 frames I P b b P b b, CQP 30, preset-medium analysis (rd 3, hex/subme 2, 3 refs), deblocking + SAO",
               "entry": "x265amd_encoder_open / x265amd_encoder_encode (include/x265amd_encoder.h)",
               "frames_per_s": len(coded) / dt, "seconds": dt, "byte_stream_md5_equals_reference_encoder": bool(same),
               "note": "one CTU at a time, every block operation a synchronous launch: latency-bound, not a throughput figure"}
        # the same with wavefront parallel processing on a 832x480 clip: one host thread + HIP stream per CTU row in flight
        g2 = np.load(os.path.join(ROOT, "tests", "golden", "encoder_api_golden.npz"))
        planes2 = T.encoder_api_clip("wvga/", 832, 480, 5)
        cfg2 = dict(fpsNum=30, fpsDenom=1, qp=30, aspectRatioIdc=1, bframes=2, bEnableLoopFilter=1, bEnableSAO=1, bEnableWavefront=1)
        t0 = time.perf_counter()
        stream2, coded2 = T.encoder_run(L, planes2, 832, 480, **cfg2)
        dt2 = time.perf_counter() - t0
        want2 = g2["wvga/stream"]
        out["wpp_832x480"] = {"clip": "832x480 8-bit, 5 frames I P b b P, same settings + WPP (13 x 8 CTUs, up to 8 CTU rows in flight)",
                              "frames_per_s": len(coded2) / dt2, "seconds": dt2,
                              "byte_stream_md5_equals_reference_encoder": bool(len(stream2) == len(want2) and hashlib.md5(stream2.tobytes()).hexdigest() == hashlib.md5(want2.tobytes()).hexdigest())}
        # and at the headline configuration's picture size (BASELINE.json configs[1] geometry): 4 frames I P b b, WPP
        planes3 = T.encoder_api_clip("fhd/", 1920, 1080, 4)
        t0 = time.perf_counter()
        stream3, coded3 = T.encoder_run(L, planes3, 1920, 1080, **cfg2)
        dt3 = time.perf_counter() - t0
        want3 = g2["fhd/stream"]
        out["wpp_1920x1080"] = {"clip": "1920x1080 8-bit, 4 frames I P b b, same settings + WPP (30 x 17 CTUs); the I frame alone takes about 6 s",
                                "frames_per_s": len(coded3) / dt3, "seconds": dt3,
                                "byte_stream_md5_equals_reference_encoder": bool(len(stream3) == len(want3) and hashlib.md5(stream3.tobytes()).hexdigest() == hashlib.md5(want3.tobytes()).hexdigest())}
        return out
    except Exception as e:     # the kernel workload above stays valid without it
        return {"error": repr(e)}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--res", choices=["1080p", "2160p"], default="1080p",
                    help="1080p = BASELINE.json configs[1] (the bench line); 2160p = the same pass at 3840x2160 (informational)")
    return ap.parse_args(argv)


def run(args, with_encoder_samples=False):
    """the kernel workload of one frame, K steps; returns the result line (rank 0) or None.  torch.distributed is the caller's business."""
    global W, H
    if args.res == "2160p":
        W, H = 3840, 2160

    import torch
    import torch.distributed as dist
    import hevc_testlib as T

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("the kernel workload needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)

    me = T.HipME(DEPTH)
    lib = me.lib
    stride = W + 2 * MARGIN_X
    origin = MARGIN_Y * stride + MARGIN_X
    nring = NUM_REFS + 6
    frames = make_clip(nring)
    # one contiguous device buffer per picture: Y | U | V  (the unit that is published to the other ranks)
    ysz, csz = frames[0][0].size, frames[0][1].size
    d_pics = [me.upload(np.concatenate([f[0].ravel(), f[1].ravel(), f[2].ravel()])) for f in frames]
    planes = [(d.data_ptr(), d.data_ptr() + ysz, d.data_ptr() + ysz + csz) for d in d_pics]

    # ---- motion estimation: identical PU set for each reference distance; planned once ----
    jobs_by_ref = [frame_jobs(T, r + 1) for r in range(NUM_REFS)]
    packed, groups, packed_unordered = [], [], []
    base = 0
    for r, jobs in enumerate(jobs_by_ref):
        pk = T.me_pack_jobs(jobs)
        packed_unordered.append(pk)
        g, order = me.plan(pk, r, MAX_WIN)
        g["first_job"] += base
        base += len(pk)
        packed.append(pk[order]); groups.append(g)
    packed = np.concatenate(packed); groups = np.concatenate(groups)
    d_groups, d_jobs = me.upload(groups), me.upload(packed)
    d_out = torch.zeros(len(packed) * 8, dtype=torch.uint8, device="cuda")
    me_bytes = int((groups["win_w"].astype(np.int64) * groups["win_h"]).sum() + len(groups) * 64 * 64 + len(packed) * (72 + 8))

    # ---- residual + intra job lists, one variant per possible current picture (addresses are absolute) ----
    curs = list(range(NUM_REFS, nring))
    tu0, arena_bytes = tu_jobs(T, planes[curs[0]], planes[curs[0] - 1], 0)
    d_arena = torch.zeros(arena_bytes, dtype=torch.uint8, device="cuda")
    d_tu, d_in, reftab = {}, {}, {}
    for cur in curs:
        tj, _ = tu_jobs(T, planes[cur], planes[cur - 1], d_arena.data_ptr())
        d_tu[cur] = me.upload(tj)
        d_in[cur] = me.upload(intra_jobs(T, planes[cur], planes[cur - 1]))
        reftab[cur] = me.upload(np.array([planes[cur - 1 - r][0] + origin for r in range(NUM_REFS)], np.uint64))
    n_tu, n_in = len(tu0), len(intra_jobs(T, planes[curs[0]], planes[curs[0] - 1]))
    d_tu_out = torch.zeros(n_tu * T.TU_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda")
    d_in_out = torch.zeros(n_in * 35, dtype=torch.int32, device="cuda")
    tu_n2 = (1 << tu0["log2"].astype(np.int64)) ** 2
    tu_bytes = int((tu_n2 * (1 + 1 + 1 + 2 + 2)).sum() + n_tu * (64 + 32))
    ij = intra_jobs(T, planes[curs[0]], planes[curs[0] - 1])
    in_n = 1 << ij["log2"].astype(np.int64)
    in_bytes = int((in_n * in_n + 4 * in_n + 1).sum() + n_in * (40 + 140))

    # ---- merge-candidate costs, intra TU steps, coefficient bits ----
    d_planetab = me.upload(np.array([[p[0] + origin, p[1] + (MARGIN_Y // 2) * (W // 2 + MARGIN_X) + MARGIN_X // 2,
                                      p[2] + (MARGIN_Y // 2) * (W // 2 + MARGIN_X) + MARGIN_X // 2] for p in planes], np.uint64).ravel())
    ic0, ic_bytes_arena = inter_cost_jobs(T, curs[0], 0)
    d_ic_arena = torch.zeros(ic_bytes_arena, dtype=torch.uint8, device="cuda")
    it0, it_bytes_arena = intra_tu_jobs(T, planes[curs[0]], planes[curs[0] - 1], 0)
    d_it_arena = torch.zeros(it_bytes_arena, dtype=torch.uint8, device="cuda")
    d_ic, d_it, d_fenctab = {}, {}, {}
    for cur in curs:
        d_ic[cur] = me.upload(inter_cost_jobs(T, cur, d_ic_arena.data_ptr())[0])
        d_it[cur] = me.upload(intra_tu_jobs(T, planes[cur], planes[cur - 1], d_it_arena.data_ptr())[0])
        d_fenctab[cur] = d_planetab.data_ptr() + cur * 24
    n_ic, n_it = len(ic0), len(it0)
    d_ic_out = torch.zeros(n_ic * 2, dtype=torch.int32, device="cuda")
    d_it_out = torch.zeros(n_it * T.TU_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda")
    ic_sz = ic0["w"].astype(np.int64) ** 2
    ic_bytes = int((ic_sz * np.where(ic0["chroma_cost"] > 0, 1.5, 1.0) * 3).sum() + n_ic * (96 + 8))     # ref read + pred write + source read
    it_n = 1 << it0["tu"]["log2"].astype(np.int64)
    it_bytes = int((it_n * it_n * (1 + 1 + 2 + 2) + 4 * it_n + 1).sum() + n_it * (96 + 32))
    ctx0 = np.zeros(160, np.uint8)
    lib.x265amd_entropy_reset(1, QP, T._ptr(ctx0))
    d_ctx0 = me.upload(ctx0)
    d_ctx_out = torch.zeros(n_tu * 160, dtype=torch.uint8, device="cuda")
    tjv = tu_jobs(T, planes[curs[0]], planes[curs[0] - 1], d_arena.data_ptr())[0]
    cb = np.zeros(n_tu, T.COEFF_BITS_JOB_DT)
    cb["coeff"], cb["ctx_in"], cb["ctx_out"] = tjv["coeff"], d_ctx0.data_ptr(), d_ctx_out.data_ptr() + np.arange(n_tu, dtype=np.int64) * 160
    cb["log2"], cb["ttype"], cb["intra"], cb["dir"], cb["signhide"] = tjv["log2"], tjv["ttype"], 0, 0, 1
    d_cb = me.upload(cb)
    d_bits = torch.zeros(n_tu, dtype=torch.int64, device="cuda")
    cb_bytes = int((tu_n2 * 2).sum() + n_tu * (32 + 8 + 320))

    # ---- in-loop filters on a synthetic reconstructed picture of the same size: random coding quad-tree (deblock), random SAO parameters ----
    hflt = (H // 8) * 8
    dbc = T.deblock_case(DEPTH, 9, W, hflt, True, False)
    sac = T.sao_case(DEPTH, 7, W, hflt)
    d_db = [me.upload(p.ravel()) for p in dbc["planes"]]
    d_db_units = me.upload(dbc["units"])
    d_sa_rec = [me.upload(p.ravel()) for p in sac["rec"]]
    d_sa_fenc = [me.upload(p.ravel()) for p in sac["fenc"]]
    d_sa_out = [torch.zeros_like(t) for t in d_sa_rec]
    satab = lambda ds: np.array([ds[0].data_ptr() + sac["org"][0], ds[1].data_ptr() + sac["org"][1], ds[2].data_ptr() + sac["org"][1]], np.uint64)
    sa_rec_tab, sa_fenc_tab, sa_out_tab = satab(d_sa_rec), satab(d_sa_fenc), satab(d_sa_out)
    d_sa_cnt = torch.zeros(sac["nctu"] * 480, dtype=torch.int32, device="cuda"); d_sa_org = torch.zeros_like(d_sa_cnt)
    d_sa_par = me.upload(sac["params"])
    pic_bytes = W * hflt * 3 // 2
    db_bytes = 2 * 2 * pic_bytes + 2 * len(dbc["units"]) * 12         # two passes, each reads and writes the picture; unit records
    ss_bytes = 2 * pic_bytes + sac["nctu"] * 480 * 8
    sap_bytes = 2 * pic_bytes + sac["nctu"] * 20
    ext_bytes = (W + 32) * (hflt + 32) * 3 // 2

    def run_filters(db_planes):
        rc = lib.x265amd_deblock_picture(sp, C.c_void_p(db_planes[0].data_ptr() + dbc["org"][0]), C.c_void_p(db_planes[1].data_ptr() + dbc["org"][1]),
                                         C.c_void_p(db_planes[2].data_ptr() + dbc["org"][1]), C.c_int64(dbc["stride"]), C.c_int64(dbc["cstride"]), W, hflt,
                                         C.c_void_p(d_db_units.data_ptr()), dbc["beta"], dbc["tc"], dbc["cb"], dbc["cr"], 0, 3)
        assert rc == 0, lib.x265amd_last_error()

    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)
    fs = _PlaneExchange
    ring, gather = fs.ReferenceRing(depth=NUM_REFS * max(world, 1) + world), [None]
    NK = 10
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(NK + 1)] for _ in range(args.steps)]

    def step(k, marks=None):
        cur = curs[k % len(curs)]
        if marks: marks[0].record(stream)
        rc = lib.x265amd_me_search(me.ctx, sp, C.c_void_p(planes[cur][0] + origin), C.c_void_p(reftab[cur].data_ptr()), C.c_int64(stride),
                                   C.c_void_p(d_groups.data_ptr()), len(groups), C.c_void_p(d_jobs.data_ptr()), C.c_void_p(d_out.data_ptr()),
                                   MAX_WIN[0], MAX_WIN[1], 0, None, C.c_int64(0))
        assert rc == 0, lib.x265amd_last_error()
        if marks: marks[1].record(stream)
        rc = lib.x265amd_intra_scan(sp, C.c_void_p(d_in[cur].data_ptr()), n_in, C.c_void_p(d_in_out.data_ptr()), None)
        assert rc == 0, lib.x265amd_last_error()
        if marks: marks[2].record(stream)
        rc = lib.x265amd_tu_chain(sp, C.c_void_p(d_tu[cur].data_ptr()), n_tu, C.c_void_p(d_tu_out.data_ptr()))
        assert rc == 0, lib.x265amd_last_error()
        if marks: marks[3].record(stream)
        rc = lib.x265amd_inter_cost(sp, C.c_void_p(d_planetab.data_ptr()), C.c_int64(stride), C.c_int64(W // 2 + MARGIN_X), W, H,
                                    C.c_void_p(d_ic[cur].data_ptr()), n_ic, C.c_void_p(d_fenctab[cur]), C.c_int64(stride), C.c_int64(W // 2 + MARGIN_X),
                                    C.c_void_p(d_ic_out.data_ptr()))
        assert rc == 0, lib.x265amd_last_error()
        if marks: marks[4].record(stream)
        rc = lib.x265amd_intra_tu_chain(sp, C.c_void_p(d_it[cur].data_ptr()), None, n_it, C.c_void_p(d_it_out.data_ptr()))
        assert rc == 0, lib.x265amd_last_error()
        if marks: marks[5].record(stream)
        rc = lib.x265amd_coeff_bits(sp, C.c_void_p(d_cb.data_ptr()), n_tu, C.c_void_p(d_bits.data_ptr()))
        assert rc == 0, lib.x265amd_last_error()
        if marks: marks[6].record(stream)
        run_filters(d_db)
        if marks: marks[7].record(stream)
        rc = lib.x265amd_sao_stats(sp, T._ptr(sa_rec_tab), T._ptr(sa_fenc_tab), C.c_int64(sac["stride"]), C.c_int64(sac["cstride"]), W, hflt,
                                   C.c_void_p(d_sa_cnt.data_ptr()), C.c_void_p(d_sa_org.data_ptr()))
        assert rc == 0, lib.x265amd_last_error()
        if marks: marks[8].record(stream)
        rc = lib.x265amd_sao_apply(sp, T._ptr(sa_rec_tab), T._ptr(sa_out_tab), C.c_int64(sac["stride"]), C.c_int64(sac["cstride"]), W, hflt, C.c_void_p(d_sa_par.data_ptr()))
        assert rc == 0, lib.x265amd_last_error()
        if marks: marks[9].record(stream)
        for c in range(3):
            rc = lib.x265amd_extend_pic_border(sp, C.c_void_p(int(sa_out_tab[c])), C.c_int64(sac["stride"] if c == 0 else sac["cstride"]),
                                               W >> (c > 0), hflt >> (c > 0), 16 >> (c > 0), 16 >> (c > 0))
            assert rc == 0, lib.x265amd_last_error()
        if marks: marks[10].record(stream)
        if world > 1:
            # exchange step of the frame-parallel design: every rank publishes the picture it just finished so that all
            # ranks hold it as a reference (here the source stands in for the reconstruction)
            gather[0] = fs.publish_step(d_pics[cur], k, ring, gather[0])

    for k in range(args.warmup):
        step(k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k, ev[k])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    kms = [float(np.mean([e[i].elapsed_time(e[i + 1]) for e in ev])) for i in range(NK)]     # HIP events on the launch stream

    if rank == 0:
        # ---- parity spot checks inside the bench: samples of this very workload (last step's frame) against the oracle ----
        orc = T.load_oracle(DEPTH)
        last = curs[(args.warmup + args.steps - 1) % len(curs)]
        res = d_out.cpu().numpy().view(T.ME_RESULT_DT)
        ok, checked, off0 = True, 0, 0
        for r in range(NUM_REFS):
            n_r = len(jobs_by_ref[r])
            pk = packed[off0:off0 + n_r]
            sel = np.arange(r, n_r, 509)
            want = T.me_run_host_batch(orc, frames[last][0].ravel(), frames[last - 1 - r][0].ravel(), stride, origin, pk[sel])
            got = np.stack([res["mv"][off0 + sel, 0], res["mv"][off0 + sel, 1], res["cost"][off0 + sel]], axis=1)
            ok &= bool(np.array_equal(want, got)); checked += len(sel)
            off0 += n_r
        # TU chain and intra scan samples
        tj = d_tu[last].cpu().numpy().view(T.TU_JOB_DT)
        tres = d_tu_out.cpu().numpy().view(T.TU_RESULT_DT)
        host = [np.concatenate([f[0].ravel(), f[1].ravel(), f[2].ravel()]) for f in (frames[last], frames[last - 1])]
        bases = (d_pics[last].data_ptr(), d_pics[last - 1].data_ptr())
        tu_ok = True
        for i in range(0, n_tu, 997):
            N = 1 << int(tj[i]["log2"])
            fo, po = int(tj[i]["fenc"]) - bases[0], int(tj[i]["pred"]) - bases[1]
            coeff = np.zeros(N * N, np.int16); resi = np.zeros((N, N), np.int16); recon = np.zeros((N, N), np.uint8); st = np.zeros(5, np.uint64)
            orc.lib.orc_tu_chain(T.off(host[0], fo), C.c_int64(int(tj[i]["fenc_stride"])), T.off(host[1], po), C.c_int64(int(tj[i]["pred_stride"])), int(tj[i]["log2"]),
                                 int(tj[i]["ttype"]), 0, 0, 1, int(tj[i]["qp"]), 1, T._ptr(coeff), T._ptr(resi), C.c_int64(N), T._ptr(recon), C.c_int64(N), T._ptr(st))
            g = tres[i]
            tu_ok &= (int(g["num_sig"]), int(g["zero_dist"]), int(g["zero_energy"]), int(g["nz_dist"]), int(g["nz_energy"])) == tuple(int(v) for v in st)
            checked += 1
        ijb = d_in[last].cpu().numpy().view(T.INTRA_JOB_DT)
        ires = d_in_out.cpu().numpy().reshape(-1, 35)
        in_ok = True
        for i in range(0, n_in, 499):
            N = 1 << int(ijb[i]["log2"])
            flags = np.array([(int(ijb[i]["avail"]) >> u) & 1 for u in range(N + 1)], np.uint8)
            rb = np.zeros(258, np.uint8); fb = np.zeros(258, np.uint8); sa = np.zeros(35, np.int32)
            orc.lib.orc_init_adi_pattern(T.off(host[1], int(ijb[i]["recon"]) - bases[1]), C.c_int64(stride), int(ijb[i]["log2"]), T._ptr(flags), 1, -1, T._ptr(rb), T._ptr(fb))
            orc.lib.orc_intra_scan(T.off(host[0], int(ijb[i]["fenc"]) - bases[0]), C.c_int64(stride), int(ijb[i]["log2"]), T._ptr(rb), T._ptr(fb), T._ptr(sa))
            in_ok &= bool(np.array_equal(sa, ires[i])); checked += 1
        # merge-candidate costs, intra TU steps and coefficient bits: samples through the oracle's batch forms on host copies
        more_ok = sample_more(T, orc, frames, last, d_ic[last], d_ic_out, d_ic_arena, d_it[last], d_it_out, d_it_arena, d_cb, d_bits, d_ctx_out,
                              d_arena, ctx0, d_pics, planes)
        checked += more_ok[1]
        # in-loop filters: whole-picture check against the oracle on fresh copies of the inputs
        d_fresh = [me.upload(p.ravel()) for p in dbc["planes"]]
        run_filters(d_fresh)
        torch.cuda.synchronize()
        want = T.deblock_run_host(orc, dbc)
        flt_ok = all(np.array_equal(d_fresh[k].cpu().numpy().view(dbc["planes"][k].dtype).reshape(dbc["planes"][k].shape), want[k]) for k in range(3))
        wcnt, worg, wout = T.sao_run_host(orc, sac)
        flt_ok &= bool(np.array_equal(d_sa_cnt.cpu().numpy(), wcnt) and np.array_equal(d_sa_org.cpu().numpy(), worg))
        for k in range(3):
            m = 16 >> (k > 0)
            wk = wout[k].copy()
            hk, wk_w = (hflt >> (k > 0)), (W >> (k > 0))
            core = wk[m:m + hk, m:m + wk_w]
            wk[:] = np.pad(core, ((m, m), (m, wk.shape[1] - wk_w - m)), mode="edge")
            got = d_sa_out[k].cpu().numpy().view(wk.dtype).reshape(wk.shape)
            flt_ok &= bool(np.array_equal(got[:, :wk_w + 2 * m], wk[:, :wk_w + 2 * m]))
        checked += 3          # three whole-picture comparisons (deblocked planes, SAO statistics, offset + extended planes)
        names = ("k_me_search", "k_intra_scan", "k_tu_chain", "k_motion_compensation<cost>", "k_intra_tu_chain", "k_coeff_bits",
                 "k_deblock<0>+<1>", "k_sao_stats", "k_sao_apply", "k_extend_border")
        algb = (me_bytes, in_bytes, tu_bytes, ic_bytes, it_bytes, cb_bytes, db_bytes, ss_bytes, sap_bytes, ext_bytes)
        dom = int(np.argmax(kms))
        line = {
            "metric": "encoded frames/sec at 1080p & 2160p --preset medium; bit-exact vs CPU ref",
            "value": world * args.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": ("%dx%d 8-bit 4:2:0 synthetic, --preset medium parameters: analysis hot-path kernels of one frame = " % (W, H)) + (
                                   "%d motion searches (every 2Nx2N PU 64..8 of every CTU x 3 refs; hex, merange 57, subme 2) + %d intra 35-mode scans "
                                   "(CUs 32/16/8) + %d TU residual chains (luma + 2 chroma per CU 32/16/8; dct, quant, sign hiding, dequant, idct, recon, sse, psy) + "
                                   "%d merge-candidate costs (2 per CU 64..8: motion compensation + SA8D incl. chroma) + %d intra TU steps (neighbours, prediction, "
                                   "residual chain; luma + 2 chroma per CU 32/16/8) + %d bits-only coefficient codings (one per TU chain) + in-loop filters of one picture "
                                   "(deblocking of a random coding quad-tree, SAO statistics, SAO application, border extension); "
                                   "the kernel work of a frame, not an encode: the real frame pipeline (mode decision + filters + bitstream, MD5-identical to the reference encoder) is reported under encoder_pipeline and is not batched across CTUs yet") % (len(packed), n_in, n_tu, n_ic, n_it, n_tu),
                       "frames_per_step_per_gpu": 1, "parallelism": "frame-per-gpu x%d" % world},
            "kernels": {names[i]: {"ms": kms[i], "algorithmic_bytes": algb[i], "GB/s": algb[i] / (kms[i] * 1e-3) / 1e9} for i in range(NK)},
            "roofline": {"bound": "hbm", "achieved": algb[dom] / (kms[dom] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": algb[dom] / (kms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": measured_traffic(names[dom]) if args.res == "1080p" else None,
                         "kernel": names[dom], "kernel_ms": kms[dom], "algorithmic_bytes_per_launch": algb[dom]},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(T, frames, packed_unordered)
        else:
            line["cpu_baseline"] = None
        line["parity_sample"] = {"checked": checked, "bit_exact_vs_oracle": bool(ok and tu_ok and in_ok and more_ok[0] and flt_ok),
                                 "me": bool(ok), "tu_chain": bool(tu_ok), "intra_scan": bool(in_ok), "inter_cost_intra_tu_coeff_bits": bool(more_ok[0]),
                                 "in_loop_filters_whole_picture": bool(flt_ok)}
        if with_encoder_samples:
            line["encoder_pipeline"] = encoder_pipeline_sample(T)
        return line
    return None


if __name__ == "__main__":
    # stand-alone: the kernel workload alone (the round-1 bench line); bench.py embeds it beside the encoder's figure
    import torch.distributed as dist
    a = parse_args()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import torch
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
    out = run(a, with_encoder_samples=False)
    if out is not None:
        print(json.dumps(out))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        dist.destroy_process_group()
