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


def lcg_noise(shape, seed):
    """integer-only noise in [-12, 12] (SURVEY.md section 8d generator)"""
    n = shape[0] * shape[1]
    idx = np.arange(n, dtype=np.uint64)
    v = (idx * np.uint64(6364136223846793005) + np.uint64(seed) * np.uint64(1442695040888963407)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    v ^= v >> np.uint64(33)
    v = (v * np.uint64(0xFF51AFD7ED558CCD)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    v ^= v >> np.uint64(29)
    return ((v % np.uint64(25)).astype(np.int64) - 12).reshape(shape)


def make_clip(nframes):
    """padded luma planes of a moving textured gradient: frame t is the base shifted by (2t, t) samples"""
    bw, bh = W + 2 * nframes + 8, H + nframes + 8
    yy, xx = np.mgrid[0:bh, 0:bw].astype(np.int64)
    base = 128 + ((xx * 3 + yy * 2) % 160 - 80) // 2 + (((xx >> 4) ^ (yy >> 4)) & 7) * 6 + lcg_noise((bh, bw), 0x9E3779B9 ^ (2 << 8))
    base = np.clip(base, 0, 255)
    frames = []
    for t in range(nframes):
        pic = base[t:t + H, 2 * t:2 * t + W] + lcg_noise((H, W), 1000 + t) // 4
        pic = np.clip(pic, 0, 255).astype(np.uint8)
        frames.append(np.pad(pic, ((MARGIN_Y, MARGIN_Y), (MARGIN_X, MARGIN_X)), mode="edge"))
    return frames


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


def cpu_baseline(T, frames, packed_by_ref, budget_s=15.0):
    """the same searches on ONE host core, timed inside one C loop: through the reference's own MotionEstimate class
    (oracle/_ref, kind "reference") when that build is present, else through the oracle port.  Bounded sample."""
    if T.have_ref():
        L, kind = T.load_ref(DEPTH), "reference"
    else:
        L, kind = T.load_oracle(DEPTH), "port"
    stride = W + 2 * MARGIN_X
    origin = MARGIN_Y * stride + MARGIN_X
    cur = frames[NUM_REFS].ravel()
    total_jobs = sum(len(j) for j in packed_by_ref)
    done, spent = 0, 0.0
    for r, pk in enumerate(packed_by_ref):
        ref = frames[NUM_REFS - 1 - r].ravel()
        for k in range(0, len(pk), 8192):
            t0 = time.perf_counter()
            T.me_run_host_batch(L, cur, ref, stride, origin, pk[k:k + 8192])
            spent += time.perf_counter() - t0
            done += len(pk[k:k + 8192])
            if spent > budget_s * (r + 1) / NUM_REFS:
                break
    fps = (done / spent) / total_jobs
    return {"value": fps, "unit": "frames/s", "cores": 1, "kind": kind,
            "sample": "%d of the %d motion searches of one 1080p frame (all PU sizes, 3 refs) in %.1f s on one core, extrapolated to frames/s"
                      % (done, total_jobs, spent)}


def measured_traffic():
    """HBM bytes per launch of the dominant kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, run
    separately: counters cannot be read inside the bench); the summary is committed under profiles/"""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_me_v2_traffic.json")) as f:
            return json.load(f)["hbm_bytes_per_launch_uncorrected"]
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import hevc_testlib as T

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    me = T.HipME(DEPTH)
    stride = W + 2 * MARGIN_X
    origin = MARGIN_Y * stride + MARGIN_X
    nring = NUM_REFS + 6
    frames = make_clip(nring)
    d_frames = [me.upload(f) for f in frames]

    # job lists: identical PU set for each reference distance; planned once (windows depend only on the predictors)
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
    alg_bytes = int((groups["win_w"].astype(np.int64) * groups["win_h"]).sum() + len(groups) * 64 * 64 + len(packed) * (72 + 8))

    stream = torch.cuda.current_stream()
    import __graft_entry__ as entry
    fs = entry.load_package().frame_shard
    ring, gather = fs.ReferenceRing(depth=NUM_REFS * max(world, 1) + world), [None]

    def step(k):
        cur = NUM_REFS + (k % (nring - NUM_REFS))
        rt = reftab[cur]
        rc = me.lib.x265amd_me_search(me.ctx, C.c_void_p(stream.cuda_stream), C.c_void_p(d_frames[cur].data_ptr() + origin),
                                      C.c_void_p(rt.data_ptr()), C.c_int64(stride), C.c_void_p(d_groups.data_ptr()), len(groups),
                                      C.c_void_p(d_jobs.data_ptr()), C.c_void_p(d_out.data_ptr()), MAX_WIN[0], MAX_WIN[1], 0)
        assert rc == 0, me.lib.x265amd_last_error()
        if world > 1:
            # exchange step of the frame-parallel design: every rank publishes the picture it just finished so that all
            # ranks hold it as a reference (here the source stands in for the reconstruction)
            gather[0] = fs.publish_step(d_frames[cur], k, ring, gather[0])

    # device tables of reference-plane addresses, one per possible current frame (built outside the timed region)
    reftab = {cur: me.upload(np.array([d_frames[cur - 1 - r].data_ptr() + origin for r in range(NUM_REFS)], np.uint64))
              for cur in range(NUM_REFS, nring)}
    for k in range(args.warmup):
        step(k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record(stream)
        step(args.warmup + k)
        ev[k][1].record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))     # HIP events on the launch stream

    if rank == 0:
        # parity spot check inside the bench: a sample of this very workload (last step's frame) against the oracle
        res = d_out.cpu().numpy().view(T.ME_RESULT_DT)
        last = NUM_REFS + ((args.warmup + args.steps - 1) % (nring - NUM_REFS))
        orc = T.load_oracle(DEPTH)
        parity_ok, checked, off0 = True, 0, 0
        for r in range(NUM_REFS):
            n_r = len(jobs_by_ref[r])
            pk = packed[off0:off0 + n_r]
            for i in range(r, n_r, 509):
                j = dict(x=int(pk[i]["x"]), y=int(pk[i]["y"]), w=int(pk[i]["w"]), h=int(pk[i]["h"]), qp=int(pk[i]["qp"]),
                         mvp=tuple(int(v) for v in pk[i]["mvp"]), mvmin=tuple(int(v) for v in pk[i]["mvmin"]), mvmax=tuple(int(v) for v in pk[i]["mvmax"]),
                         mvc=[tuple(int(v) for v in pk[i]["mvc"][k]) for k in range(int(pk[i]["num_cand"]))], merange=int(pk[i]["merange"]),
                         method=int(pk[i]["method"]), subme=int(pk[i]["subme"]))
                want = T.me_run_host(orc, frames[last].ravel(), frames[last - 1 - r].ravel(), stride, origin, [j])[0]
                got = (int(res[off0 + i]["mv"][0]), int(res[off0 + i]["mv"][1]), int(res[off0 + i]["cost"]))
                parity_ok &= tuple(int(v) for v in want) == got
                checked += 1
            off0 += n_r
        line = {
            "metric": "encoded frames/sec at 1080p & 2160p --preset medium; bit-exact vs CPU ref",
            "value": world * args.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "1920x1080 8-bit 4:2:0 synthetic, --preset medium search parameters (hex, merange 57, subme 2, 3 refs): "
                                   "motion-estimation hot path only (every 2Nx2N PU 64..8 of every CTU x 3 refs = %d searches/frame); "
                                   "NOT a full encode" % len(packed),
                       "frames_per_step_per_gpu": 1, "searches_per_frame": int(len(packed)), "parallelism": "frame-per-gpu x%d" % world},
            "roofline": {"bound": "hbm", "achieved": alg_bytes / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg_bytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": measured_traffic(),
                         "kernel": "k_me_search", "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": alg_bytes},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(T, frames, packed_unordered)
        else:
            line["cpu_baseline"] = None
        line["parity_sample"] = {"checked": checked, "bit_exact_vs_oracle": bool(parity_ok)}
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
