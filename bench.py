#!/usr/bin/env python3
"""bench.py -- encoded frames per second of the MI355X HEVC encoder object on BASELINE.json's metric / configs[1].

A "step" is ONE ENCODED FRAME of a synthetic 1920x1080 8-bit 4:2:0 clip through the drop-in boundary (include/x265amd_encoder.h:
x265amd_encoder_open / encode / close, the reference's x265_encoder_* entry points, source/x265.h:2412-2471): mode decision of every CTU
(Analysis::compressCTU with the --preset medium analysis settings: rd 3, hex / subme 2, 3 references, early skip, rskip, psy-rd 2, sign hiding,
4 B frames), in-loop deblocking + SAO, CABAC, slice NAL units -- the whole byte stream, which must equal the reference ENCODER's (oracle/_ref/x265_ref8,
the reference compiled from /root/reference; same clip, same options) byte for byte.  That comparison is made in the run and reported; the
reference's own wall time on the host cores of the same box is the CPU baseline (kind "reference").

Timed region: W warm-up frames are encoded (untimed, their own encoder object), then EXACTLY K frames between barriers: from the first
x265amd_encoder_encode of the K-frame clip to the last flushed NAL unit.  The boundary takes host pictures (as x265_picture does): each input frame
is uploaded inside the timed region (3.1 MB, about 60 us over PCIe against > 100 ms of analysis).

Multi-GPU (one process per GPU, torch.distributed nccl == RCCL), two ways:
  default (--shard frames)  SURVEY section 8e as written: ONE clip, picture k in coding order coded by rank k mod N, every finished CTU row published to the other ranks
                  (x265amd_encoder_export_row / _import_row, pump: x265-amod_amd/frame_rows.py, ncclBroadcast); total work is fixed: "scaling": "strong";
  --shard gops    rank r encodes closed GOP r of the clip (keyint K, x265amd_param.firstFrame = r K), no data-path collective, the coded GOPs' sizes and digests
                  are gathered for the report; per-GPU work is fixed: "scaling": "weak".
The GOP structure is the lookahead's (--b-adapt 2, B pyramid, open GOPs, scene-cut detection: the preset as it comes), not fixed mini-GOPs.

The kernels of the hot path are timed on their own in bench_kernels.py (a frame's worth of motion searches, intra scans, transform chains, merge
costs, coefficient codings and filters as batches): its figures ride along as `kernel_workload`.  The `roofline` object prices the kernel the timed
region actually runs: k_job_server, resident for the whole encode -- the algorithmic bytes of every command it ran (counted on the device) over its launch
duration (HIP events on its stream), and the share of its resident time spent inside command bodies (`busy_frac`).

Contract: python bench.py --gpus N --steps K --warmup W ; prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H = 1920, 1080
HBM_PEAK_GBS = 8000.0                # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BFRAMES, REFS, CRF = 4, 3, 28

# x265amd_param fields that differ from x265amd_param_default, and the same settings on the reference's command line
# frameNumThreads > 1: the reference's frame-parallel rules, i.e. what its default (--frame-threads 0 = by core count) gives on any machine with four cores or more
ENC_CFG = dict(fpsNum=30, fpsDenom=1, aspectRatioIdc=1, bframes=BFRAMES, bEnableLoopFilter=1, bEnableSAO=1, bEnableWavefront=1, frameNumThreads=5,
               scenecutThreshold=40, lookaheadDepth=20, bFrameAdaptive=2, bOpenGOP=1, bBPyramid=1, lookaheadSlices=8, bEnableWeightedPred=1,
               rateControlMode=2, rfConstant=float(CRF), aqMode=2, aqStrength=1.0, cuTree=1, qCompress=0.6, qgSize=32)
# Every value of ENC_CFG is the preset's, its rate control included (round 6): the reference runs plain `--preset medium` -- constant rate factor 28, --aq-mode 2, cuTree
# (source/common/param.cpp:266-290) -- which is the command line BASELINE.json's metric names; --no-info leaves out the SEI NAL unit that carries the reference build's version
# and option string
REF_CLI = ["--preset", "medium", "--no-info"]


def bench_clip(first, count, gop=0, depth=8, cfg_id=2):
    """frames first .. first + count - 1 (display order) of the synthetic clip SURVEY.md section 8d prescribes (tests/hevc_testlib.py: survey_clip, cfg_id 2 --
    the generator the full-size parity cases use as well): integer arithmetic only, frame t depends on t alone, the noise field is re-seeded every 24th frame.
    gop: the closed GOP a rank codes in a multi-GPU run (its own noise field behind its IDR picture, the same motion: every rank has the same amount of work)."""
    import hevc_testlib as T
    return T.survey_clip(W, H, depth, cfg_id, first, count, gop)


def encode(T, L, frames, first_frame, keyint, sync, timed=True, shard=None, cfg=None, host_input=False):
    """the clip through the encoder object; returns (byte stream, seconds of the encode loop).  `sync` brackets the timed region.
    The frames are on the GPU before the timed region starts and go in through x265amd_encoder_encode_device (include/x265amd_encoder.h); host_input: they stay host
    arrays and go in through x265amd_encoder_encode, padded on the host and sent over the bus inside the timed region (the reference API's form: reported beside the value).
    shard = (rank, world): frame per GPU with row publication -- this object codes the pictures whose place in coding order is rank modulo
    world, a pump thread beside the encode loop broadcasts / imports every finished CTU row (x265-amod_amd/frame_rows.py); the stream holds the owned pictures only."""
    lib = L.lib
    lib.x265amd_encoder_open.restype = C.c_void_p
    lib.x265amd_encoder_open.argtypes = [C.POINTER(T.EncParam)]
    lib.x265amd_encoder_headers.argtypes = [C.c_void_p, C.POINTER(C.POINTER(T.EncNal)), C.POINTER(C.c_uint32)]
    lib.x265amd_encoder_encode.argtypes = [C.c_void_p, C.POINTER(C.POINTER(T.EncNal)), C.POINTER(C.c_uint32), C.POINTER(T.EncPicture), C.POINTER(T.EncPicture)]
    lib.x265amd_encoder_encode_device.argtypes = lib.x265amd_encoder_encode.argtypes
    lib.x265amd_encoder_close.argtypes = [C.c_void_p]
    lib.x265amd_param_default.argtypes = [C.POINTER(T.EncParam)]
    lib.x265amd_last_error.restype = C.c_char_p
    encode_fn = lib.x265amd_encoder_encode if host_input else lib.x265amd_encoder_encode_device
    prm = T.EncParam()
    lib.x265amd_param_default(C.byref(prm))
    prm.sourceWidth, prm.sourceHeight = W, H
    for k, v in (cfg or ENC_CFG).items():
        setattr(prm, k, v)
    prm.firstFrame = first_frame
    if keyint:
        prm.keyframeMax = keyint
    if shard:
        prm.shardRank, prm.shardCount = shard[0], shard[1]
    enc = lib.x265amd_encoder_open(C.byref(prm))
    if not enc:
        raise SystemExit("x265amd_encoder_open: %s" % lib.x265amd_last_error().decode())
    stream = bytearray()
    nal = C.POINTER(T.EncNal)(); nnal = C.c_uint32(0)
    if first_frame == 0:
        assert lib.x265amd_encoder_headers(enc, C.byref(nal), C.byref(nnal)) > 0
        for i in range(nnal.value):
            stream += bytes(nal[i].payload[:nal[i].sizeBytes])
    pics = []
    for planes in frames:
        pic = T.EncPicture()
        keep = [np.ascontiguousarray(pl) for pl in planes]
        if host_input:
            for k in range(3):
                pic.planes[k] = keep[k].ctypes.data; pic.stride[k] = keep[k].strides[0]
        else:
            import torch
            keep = [torch.from_numpy(k.view(np.uint8).reshape(k.shape[0], -1)).cuda() for k in keep]         # (bytes: torch has no unsigned 16-bit type)
            for k in range(3):
                pic.planes[k] = keep[k].data_ptr(); pic.stride[k] = keep[k].stride(0)
        pics.append((pic, keep))
    coded = 0
    marks = []                  # where each coded picture's NAL units start in `stream` (the headers come first)
    lib.x265amd_encoder_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
    try:
        if timed:
            sync()
        pump_thread, pump_err = None, []
        if shard:
            import threading
            import torch
            import __graft_entry__ as g
            fr = g.load_package().frame_rows
            dev = "cuda:%d" % torch.cuda.current_device()       # read on the encode thread: a new thread starts on device 0

            def run_pump():
                try:
                    torch.cuda.set_device(dev)
                    rows = fr.EncoderRows(lib, enc, dev)
                    # one pump thread on a stream of its own, the same sequence of collectives on every rank, pictures nobody references stay home (frame_rows.py)
                    fr.pump(rows.export_row, rows.import_row, rows.shapes, len(pics), rows.rows, dev, rank=shard[0], world=shard[1], referenced=rows.referenced)
                except BaseException as exc:        # noqa: B902
                    import traceback
                    traceback.print_exc()           # at once: the encode loop may sit in the encoder for minutes before anybody asks for pump_err
                    pump_err.append(repr(exc))
            pump_thread = threading.Thread(target=run_pump)
        t0 = time.perf_counter()
        if pump_thread:
            pump_thread.start()
        for pic, _ in pics:
            ret = encode_fn(enc, C.byref(nal), C.byref(nnal), C.byref(pic), None)
            assert ret >= 0, lib.x265amd_last_error()
            if ret:
                coded += 1
                marks.append(len(stream))
                for i in range(nnal.value):
                    stream.extend(bytes(nal[i].payload[:nal[i].sizeBytes]))
        while True:
            ret = encode_fn(enc, C.byref(nal), C.byref(nnal), None, None)
            assert ret >= 0, lib.x265amd_last_error()
            if not ret:
                break
            coded += 1
            marks.append(len(stream))
            for i in range(nnal.value):
                stream.extend(bytes(nal[i].payload[:nal[i].sizeBytes]))
        if pump_thread:
            pump_thread.join()
            if pump_err:
                raise SystemExit("the row pump failed: %s" % pump_err[0])
        if timed:
            sync()
        dt = time.perf_counter() - t0
        st = (C.c_uint64 * 4)()
        encode.last_stats = list(st) if lib.x265amd_encoder_stats(enc, st, 4) == 0 else None      # I / P / B pictures, the sum of their distinct reference pictures
    finally:
        lib.x265amd_encoder_close(enc)
    encode.last_marks = marks           # (a sharded object returns every picture, in coding order; the pictures it does not code come without NAL units)
    assert coded == len(frames), (coded, len(frames))
    return bytes(stream), dt


def nal_count(stream, nal_type):
    """NAL units of one type in an Annex-B stream"""
    n, i = 0, 0
    while True:
        i = stream.find(b"\x00\x00\x01", i)
        if i < 0 or i + 3 >= len(stream):
            return n
        n += ((stream[i + 3] >> 1) & 63) == nal_type
        i += 3


def queue_stats(L, reset):
    """x265amd_queue_stats (include/x265amd.h): the counters of the resident kernel k_job_server since the last reset"""
    out = (C.c_uint64 * 106)()
    L.lib.x265amd_queue_stats.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.c_int]
    rc = L.lib.x265amd_queue_stats(out, 106, 1 if reset else 0)
    return list(out) if rc == 0 else None


XA_OPS = ["nop/fence", "exit", "copy", "copy2d", "fill", "copy_rects", "mc", "mc_cost", "cu_measure", "tu_chain", "tu_chain_rdoq", "intra_tu_chain", "intra_tu_chain_rdoq", "intra_scan",
          "me_search", "me_search_star", "me_deferred", "est_bit", "intra_pu", "intra_nxn", "inter_chain", "inter_search"]


def job_server_roofline(st, frames_payload_bytes, wall_s):
    """the roofline object of the timed region's dominant (and only resident) kernel.  One launch of k_job_server lasts the whole encode.  `achieved` = SURVEY section
    8d's algorithmic bytes of that launch (frame payload x (source + reconstruction + R reference pictures), R from the encoder) / the launch duration by HIP events on
    its stream; `per_command_bytes` = the sum over its commands of what each has to read and write (counted on the device) over the same duration; busy_frac = ticks
    inside command bodies / resident ticks summed over the workgroups: the rest is polling for the next command."""
    if not st or not st[6]:
        return None
    kernel_s = st[7] / 1e6
    per_op = {XA_OPS[k]: {"commands": st[10 + 3 * k], "body_ms": st[11 + 3 * k] / 1e5, "algorithmic_MB": st[12 + 3 * k] / 1e6} for k in range(len(XA_OPS)) if st[10 + 3 * k]}
    launch_s = kernel_s / st[6]
    achieved = frames_payload_bytes / launch_s / 1e9
    per_cmd = st[4] / kernel_s / 1e9
    return {"bound": "hbm", "kernel": "k_job_server", "launches": st[6], "workgroups": st[8], "launch_ms": 1000.0 * launch_s,
            "bytes": frames_payload_bytes, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "of": "SURVEY section 8d's algorithmic bytes of the launch -- frame payload x (source read + reconstruction write + R distinct reference pictures read), R per picture "
                  "as the encoder's own reference lists have it (x265amd_encoder_stats) -- over the launch's duration by HIP events on its stream (one launch = the timed encode)",
            "busy_frac": st[1] / st[5] if st[5] else None, "commands": st[0], "body_ms_all_workgroups": st[1] / 1e5, "polling_ms_all_workgroups": st[3] / 1e5,
            "per_command_bytes": {"bytes": st[4], "achieved": per_cmd, "frac": per_cmd / HBM_PEAK_GBS,
                                  "of": "the sum over the launch's commands of what each has to read and write once (counted on the device from its job records): blocks are read "
                                        "again by every command that evaluates them, so this is several times the 8d figure"},
            "per_command_kind": per_op,
            "per_command_kind_traffic": command_traffic(),
            "note": "k_job_server is resident for the whole encode; the encoder is bound by the latency of the reference's serial decision chain (busy_frac: the share of the "
                    "workgroups' resident time inside command bodies), not by bandwidth.  traffic: the PMC counter passes serialise dispatches, which a resident kernel does not "
                    "survive, so the launch itself cannot be counted; per_command_kind_traffic holds what the counters saw of its command kinds when five pictures of this clip "
                    "were coded with every command as an ordinary launch of the same device code (profiles/collect_traffic.sh 5, profiles/r06_p_pictures_traffic.json)"}


def command_traffic():
    """profiles/r06_p_pictures_traffic.json (I P B B B of this clip; round 4: the I picture alone) reduced to {command kind: launches, HBM-side bytes by the counters (low: FETCH_SIZE +
    WRITE_SIZE as reported; high: FETCH_SIZE doubled, the guide's gfx950 correction for wide requests), algorithmic bytes, ratio of the totals, and -- because with the job
    server off the P / B pictures issue other NUMBERS of units per kind -- a launch's traffic against a command's own algorithmic bytes}: measured once on a GPU box with every
    command an ordinary launch of the same device code (profiles/collect_traffic.sh 5), not during this run.  The skip chain and the fused search exist as job-server commands
    only and cannot be launched for the counters."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r06_p_pictures_traffic.json")))
    except (OSError, ValueError):
        return None
    out = {}
    for name, r in t.get("kernels", {}).items():
        if "command_kind" in r:
            out[r["command_kind"]] = {k: r[k] for k in ("launches", "hbm_bytes_low", "hbm_bytes_high", "algorithmic_bytes", "traffic_over_algorithmic_low", "traffic_over_algorithmic_high",
                                                        "hbm_bytes_per_launch_low", "hbm_bytes_per_launch_high", "algorithmic_bytes_per_command",
                                                        "per_unit_traffic_over_algorithmic_low", "per_unit_traffic_over_algorithmic_high") if k in r}
    return out or None


def usable_cores():
    """the host cores this process may really use: the cgroup's CPU quota when there is one (the GPU boxes of this project: 16 of 256 hardware threads)"""
    n = os.cpu_count() or 1
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def reference_encode(frames, cli=None, depth=8, runs=("default", "f1")):
    """the same clip through the reference encoder (oracle/_ref/x265_ref8: the reference compiled by oracle/build_ref.sh, C primitives, no assembly) on
    this box's host cores, twice: with its defaults (frame threads by core count -- 5 for 1080p on 32 cores or more, threadpool.cpp:661-677: THAT stream is
    the parity target, the encoder object runs with the same frame-parallel rules) and with --frame-threads 1 (informational: one picture at a time; vertical
    motion is then not limited to the lag, so a clip with enough motion codes differently).  Returns a dict or None."""
    exe = os.path.join(ROOT, "oracle", "_ref", "x265_ref%d" % depth)
    if not os.path.exists(exe):
        return None
    d = tempfile.mkdtemp(prefix="x265amd_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        with open(os.path.join(d, "clip.y4m"), "wb") as f:
            f.write(b"YUV4MPEG2 W%d H%d F30:1 Ip A1:1 C420%s\n" % (W, H, b"" if depth == 8 else b"p%d" % depth))
            for fr in frames:
                f.write(b"FRAME\n")
                for pl in fr:
                    f.write(np.ascontiguousarray(pl).tobytes())
        out = {"cores": usable_cores()}
        # pools16: the thread pool sized by the CPU quota of the GPU boxes (16 cores) instead of the machine's 256 hardware threads -- a process that burns more than
        # its quota is frozen for the rest of every 100 ms period -- with the frame threads the default run picks (so the rules, and the stream, are the same)
        ft = 6 if H > 2000 else 5
        variants = {"default": [], "f1": ["--frame-threads", "1"], "pools16": ["--pools", "16", "--frame-threads", str(ft if (os.cpu_count() or 1) >= 32 else 0)]}
        for tag in runs:
            extra = variants[tag]
            t0 = time.perf_counter()
            r = subprocess.run([exe, "--input", "clip.y4m", "-o", "out.hevc"] + (cli or REF_CLI) + (["--input-depth", str(depth), "--output-depth", str(depth)] if depth != 8 else []) + extra,
                               cwd=d, capture_output=True, text=True, timeout=900)
            wall = time.perf_counter() - t0
            if r.returncode != 0:
                return None
            own = None          # the encoder's own figure excludes start-up and reading the file: "encoded N frames in H:MM:SS.ss (F fps)"
            tail = r.stderr.strip().splitlines()[-1] if r.stderr.strip() else ""
            if tail.startswith("encoded ") and " fps)" in tail:
                try:
                    own = len(frames) / float(tail.split("(")[1].split(" fps")[0])
                except (IndexError, ValueError, ZeroDivisionError):
                    own = None
            out[tag] = {"stream": open(os.path.join(d, "out.hevc"), "rb").read(), "seconds": own if own else wall, "wall": wall, "says": tail}
        return out
    finally:
        for n in os.listdir(d):
            os.unlink(os.path.join(d, n))
        os.rmdir(d)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24, help="frames in the timed encode")
    ap.add_argument("--warmup", type=int, default=4, help="frames of the untimed warm-up encode")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the reference encoder's run (profiling passes)")
    ap.add_argument("--no-kernel-workload", action="store_true", help="skip bench_kernels.py (profiling passes of the encoder alone)")
    ap.add_argument("--res", choices=["1080p", "2160p"], default="1080p", help="1080p = BASELINE.json configs[1] (the bench line); 2160p: the same encode at 3840x2160 (informational)")
    ap.add_argument("--no-2160p", action="store_true", help="skip the 3840x2160 encode that the 1080p single-GPU run reports beside the bench line (`also_2160p`)")
    ap.add_argument("--shard", choices=["gops", "frames"], default="frames", help="--gpus N > 1: frames (the default: SURVEY section 8e as written, DESIGN.md section 6) = ONE clip, picture k in "
                    "coding order coded by rank k mod N, finished CTU rows broadcast over RCCL (strong scaling); gops = a closed GOP per GPU (weak scaling, no data-path exchange)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="torch.distributed backend of a multi-process run: nccl = RCCL over xGMI; gloo stages the rows through "
                    "the host (the one-GPU test of the --shard frames leg: RCCL refuses two ranks on one device)")
    ap.add_argument("--one-gpu", action="store_true", help="every rank uses cuda:0 (tests/test_encoder_api.py: two processes, one GPU, gloo); set X265AMD_QUEUES so that the ranks' resident workgroups fit side by side")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the 3840x2160 --preset slow and Main 10 encodes that the 1080p single-GPU run reports beside the bench line")
    ap.add_argument("--no-8k", action="store_true", help="skip the three 7680x4320 --preset veryslow --rd 6 frames (BASELINE configs[4]) among the extra configurations: they add about two minutes")
    ap.add_argument("--no-scene-clip", action="store_true", help="skip the 60-frame clip with both re-seeds inside that the 1080p single-GPU run reports beside the bench line (`scene_change_clip`)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # run plainly with --gpus N: start the N ranks (one process per GPU) BEFORE anything here touches the GPU, relay rank 0's line, leave with the launcher's code --
        # never a one-GPU run that calls itself N GPUs
        import socket
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist
    import hevc_testlib as T
    global W, H
    if args.res == "2160p":
        W, H = 3840, 2160

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not args.one_gpu:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if args.one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    L = T.load_hip(8)
    K, Wm = args.steps, args.warmup
    # --shard gops: rank r codes GOP r, frames r K .. r K + K - 1 of the clip, IDR first (closed GOPs need nothing from each other);
    # --shard frames: every rank is fed the same K frames and codes the pictures whose place in coding order is its rank modulo the world
    by_frames = args.shard == "frames"
    frames = bench_clip(0, K, gop=0 if by_frames else rank)
    if Wm > 0:
        encode(T, L, bench_clip(0, Wm), 0, 0, sync, timed=False)
    queue_stats(L, True)                 # the counters of the resident kernel from here on: the timed encode alone
    if by_frames:
        if world > 1:
            # the communicator is brought up before the timed region (the first collective of a process group builds its rings)
            t_ = torch.zeros(world, 4, dtype=torch.int64, device="cuda")
            dist.all_reduce(t_)
            dist.broadcast(torch.zeros(16, dtype=torch.uint8, device="cuda"), src=0)
            torch.cuda.synchronize()
        stream, dt = encode(T, L, frames, 0, 0, sync, shard=(rank, world) if world > 1 else None)
    else:
        stream, dt = encode(T, L, frames, rank * K, K if world > 1 else 0, sync)
    qstats = queue_stats(L, True)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        sizes = [None] * world
        dist.all_gather_object(sizes, (len(stream), hashlib.md5(stream).hexdigest()))
        if by_frames and os.environ.get("X265AMD_BENCH_STREAM_OUT"):
            # the one-GPU test of this leg puts the ranks' NAL units back together and compares with the single-object stream
            with open(os.environ["X265AMD_BENCH_STREAM_OUT"] + ".%d" % rank, "wb") as f_:
                f_.write(stream)
            with open(os.environ["X265AMD_BENCH_STREAM_OUT"] + ".%d.marks" % rank, "w") as f_:
                json.dump(encode.last_marks, f_)
    else:
        sizes = [(len(stream), hashlib.md5(stream).hexdigest())]

    line = None
    if rank == 0:
        ref = None if (world > 1 or args.no_cpu_baseline) else reference_encode(frames, runs=("default", "f1", "pools16"))
        if ref is not None:
            same = ref["default"]["stream"] == stream
            cpu = {"value": K / ref["default"]["seconds"], "unit": "frames/s", "cores": ref["cores"], "kind": "reference",
                   "sample": "the same %d-frame clip and options through oracle/_ref/x265_ref8 (the reference itself compiled from /root/reference, C primitives: no "
                             "assembler in the image) on this box's host cores (%d usable: the cgroup's CPU quota; the machine has %d hardware threads), thread pool and "
                             "frame threads at their defaults; by its own 'encoded N frames in T' figure" % (K, ref["cores"], os.cpu_count() or 0),
                   "frame_threads_default": {"frames_per_s": K / ref["default"]["seconds"], "says": ref["default"]["says"], "stream_equals_ours": bool(same)},
                   "frame_threads_1": {"frames_per_s": K / ref["f1"]["seconds"], "says": ref["f1"]["says"],
                                       "stream_equals_default": bool(ref["default"]["stream"] == ref["f1"]["stream"])},
                   "pools16": {"frames_per_s": K / ref["pools16"]["seconds"], "says": ref["pools16"]["says"], "stream_equals_default": bool(ref["default"]["stream"] == ref["pools16"]["stream"]),
                               "of": "the reference with --pools 16 (its thread pool sized by this box's CPU quota instead of the machine's hardware threads; frame threads as the default run picks them): "
                                     "the fairer of the two baselines where a quota is in force"}}
        else:
            same, cpu = None, None
        # algorithmic bytes of a frame = payload x (source read + reconstruction write + distinct reference pictures read)
        payload = W * H * 3 // 2
        n_i = 1
        n_b = nal_count(stream, 0) + nal_count(stream, 8)                                                          # TRAIL_N / RASL_N slices: the B pictures the lookahead chose (--b-adapt 2)
        n_p = K - n_i - n_b
        es = getattr(encode, "last_stats", None)
        if es:
            alg = payload * (2 * (es[0] + es[1] + es[2]) + es[3])                             # R per picture from the encoder's own reference lists
        else:
            alg = payload * (2 * n_i + (2 + min(REFS, 2)) * n_p + 4 * n_b)
        line = {
            "metric": "encoded frames/sec at 1080p & 2160p --preset medium; bit-exact vs CPU ref",
            "value": (K if by_frames else world * K) / dt, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": 1000.0 * dt / K,
            "higher_is_better": True, "scaling": "strong" if by_frames else "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%dx%d 8-bit 4:2:0 synthetic clip, %d frames per GPU (I + mini-GOPs of up to %d B frames chosen by the lookahead's trellis, --b-adapt 2), encoded END TO END by the encoder object "
                                   "(x265amd_encoder_open / encode / close): --preset medium analysis settings (CTU 64, rd 3, hex / merange 57 / subme 2, %d references, "
                                   "3 merge candidates, early skip, rskip, psy-rd 2.0, sign hiding, TU depth 1), deblocking, SAO, WPP, frame-parallel rules (the reference's default frame threads), CABAC, Annex-B stream; the preset's own rate "
                                   "control -- constant rate factor %d, adaptive quantisation (--aq-mode 2), cuTree, a QP per 32x32 quantisation group (cu_qp_delta) -- with "
                                   "scene-cut detection (--scenecut 40, --rc-lookahead 20), the B-frame trellis (--b-adapt 2), the B pyramid, open GOPs, the lookahead's slices (--lookahead-slices 8) and weighted prediction (--weightp: the analysis runs, no weights are chosen on these clips; "
                                   "clips with fades are coded with weights in tests/test_encoder_api.py) as the preset has them: the reference runs plain --preset medium" % (W, H, K, BFRAMES, REFS, CRF),
                       "frames_per_step_per_gpu": 1, "parallelism": ("picture k in coding order on GPU k mod %d, CTU rows broadcast over RCCL" % world if by_frames else "closed GOP per GPU x%d" % world) if world > 1 else "one encoder object",
                       "reference_command_line": "x265 --input clip.y4m -o out.hevc " + " ".join(REF_CLI)},
            "bit_exact_vs_reference_encoder": same,
            "stream": {"bytes_per_gop": [s[0] for s in sizes], "md5_per_gop": [s[1] for s in sizes]},
            "cpu_baseline": cpu,
            "roofline": job_server_roofline(qstats, alg, dt),
            "encoder_hbm": {"algorithmic_bytes": alg, "achieved_GB/s": alg / dt / 1e9, "frac_of_hbm_peak": alg / dt / 1e9 / HBM_PEAK_GBS,
                            "note": "SURVEY 8d's end-to-end figure: frame payload x (source read + reconstruction write + reference pictures read); the encoder is bound "
                                    "by the latency of the reference's serial decision chain, not by bandwidth"},
        }
    # ---- the same encode with the frames handed over as HOST buffers (the reference API's form): padded on the host and sent over the bus inside the timed region ----
    if rank == 0 and world == 1:
        try:
            stream_h, dt_h = encode(T, L, frames, 0, 0, sync, host_input=True)
            line["host_input"] = {"value": K / dt_h, "unit": "frames/s", "stream_equals": bool(stream_h == stream),
                                  "note": "x265amd_encoder_encode with x265_picture-style host planes: the PCIe-inclusive rate; `value` is x265amd_encoder_encode_device with the frames in HBM before the timed region"}
        except Exception as exc:         # noqa: BLE001
            line["host_input"] = {"error": repr(exc)}
    # ---- SURVEY.md section 8d's whole clip: 60 frames with both re-seeds of the noise field inside (pictures with new content everywhere) ----
    if rank == 0 and world == 1 and args.res == "1080p" and not args.no_scene_clip and not args.no_cpu_baseline:
        try:
            frames60 = bench_clip(0, 60)
            stream60, dt60 = encode(T, L, frames60, 0, 0, sync)
            ref60 = reference_encode(frames60, runs=("default", "f1", "pools16"))
            line["scene_change_clip"] = {"value": 60 / dt60, "unit": "frames/s", "frames": 60, "stream_md5": hashlib.md5(stream60).hexdigest(),
                                         "bit_exact_vs_reference_encoder": None if ref60 is None else bool(ref60["default"]["stream"] == stream60),
                                         "cpu_baseline": None if ref60 is None else {"value": 60 / ref60["default"]["seconds"], "cores": ref60["cores"], "kind": "reference",
                                                                                     "says": ref60["default"]["says"], "frame_threads_1": 60 / ref60["f1"]["seconds"], "pools16": 60 / ref60["pools16"]["seconds"]},
                                         "note": "the same clip generator and options over 60 frames: the noise field is re-seeded at frames 24 and 48; both encoders run "
                                                 "scene-cut detection (--scenecut 40, --rc-lookahead 20) and place an I picture at 24 (min-keyint not reached) and an IDR picture at 48"}
            del frames60, stream60
        except Exception as exc:       # the bench line stands on its own
            line["scene_change_clip"] = {"error": repr(exc)}
        # ---- and a LONG clip of the same generator (a scene cut's I picture every 24 frames): what an encode that is not over after one I picture gets ----
        try:
            n_long = 240
            framesL = bench_clip(0, n_long)
            streamL, dtL = encode(T, L, framesL, 0, 0, sync)
            refL = reference_encode(framesL, runs=("default", "pools16"))
            line["long_clip"] = {"value": n_long / dtL, "unit": "frames/s", "frames": n_long, "stream_md5": hashlib.md5(streamL).hexdigest(),
                                 "bit_exact_vs_reference_encoder": None if refL is None else bool(refL["default"]["stream"] == streamL),
                                 "cpu_baseline": None if refL is None else {"value": n_long / refL["default"]["seconds"], "cores": refL["cores"], "kind": "reference",
                                                                            "says": refL["default"]["says"], "pools16": n_long / refL["pools16"]["seconds"]},
                                 "note": "the same clip generator and options over %d frames at 1920x1080: the noise field is re-seeded every 24th frame, both encoders place an I picture there" % n_long}
            del framesL, streamL
        except Exception as exc:
            line["long_clip"] = {"error": repr(exc)}
    # ---- the same encode at 3840x2160 (BASELINE.json's metric names both sizes; the bench line is the 1080p one): reported beside it, never instead of it ----
    if rank == 0 and world == 1 and args.res == "1080p" and not args.no_2160p and not args.no_cpu_baseline:
        try:
            W, H = 3840, 2160
            frames4 = bench_clip(0, K)
            encode(T, L, bench_clip(0, 2), 0, 0, sync, timed=False)
            stream4, dt4 = encode(T, L, frames4, 0, 0, sync)
            ref4 = reference_encode(frames4, runs=("default", "f1", "pools16"))
            line["also_2160p"] = {"value": K / dt4, "unit": "frames/s", "frames": K, "stream_md5": hashlib.md5(stream4).hexdigest(),
                                  "bit_exact_vs_reference_encoder": None if ref4 is None else bool(ref4["default"]["stream"] == stream4),
                                  "cpu_baseline": None if ref4 is None else {"value": K / ref4["default"]["seconds"], "cores": ref4["cores"], "kind": "reference",
                                                                             "says": ref4["default"]["says"], "frame_threads_1": K / ref4["f1"]["seconds"], "pools16": K / ref4["pools16"]["seconds"]},
                                  "note": "the same clip generator, options and comparison at 3840x2160 8-bit (34 CTU rows)"}
            del frames4, stream4
        except Exception as exc:       # the bench line stands on its own
            line["also_2160p"] = {"error": repr(exc)}
        finally:
            W, H = 1920, 1080
    # ---- BASELINE.json configs[2] and configs[3] at their stated size: 3840x2160 --preset slow (8-bit) and 3840x2160 Main 10 --preset medium, stream compared in the run ----
    if rank == 0 and world == 1 and args.res == "1080p" and not args.no_extra_configs and not args.no_cpu_baseline:
        # (configs[4], 7680x4320 10-bit --preset veryslow --rd 6, rides along since round 5's end with three frames -- I, P and a B picture: it goes CU by CU through the host's path,
        #  what the line says is where that stands against the reference, not a figure anybody has worked on)
        extra = (("also_2160p_slow", 8, 3, T.SLOW_TOOLS, ["--preset", "slow"], (3840, 2160), 12, 2, 2),
                 ("also_2160p_main10", 10, 4, {}, ["--preset", "medium"], (3840, 2160), 12, 2, 3),
                 ("also_4320p_veryslow_rd6", 10, 5, dict(T.VERYSLOW_TOOLS, **T.VERYSLOW_GOP), ["--preset", "veryslow", "--rd", "6"], (7680, 4320), 3, 1, 4))
        for key, depth, cfg_id, tools, presetCli, size, kmax, kwarm, cfgIndex in extra:
            if key == "also_4320p_veryslow_rd6" and args.no_8k:
                continue
            try:
                W, H = size
                Kx = min(K, kmax)
                Lx = T.load_hip(depth)
                cfgx = dict(ENC_CFG, frameNumThreads=6, **tools)
                framesx = bench_clip(0, Kx, depth=depth, cfg_id=cfg_id)
                encode(T, Lx, bench_clip(0, kwarm, depth=depth, cfg_id=cfg_id), 0, 0, sync, timed=False, cfg=cfgx)
                streamx, dtx = encode(T, Lx, framesx, 0, 0, sync, cfg=cfgx)
                clix = presetCli + ["--no-info"]
                refx = reference_encode(framesx, cli=clix, depth=depth, runs=("default", "pools16"))
                line[key] = {"value": Kx / dtx, "unit": "frames/s", "frames": Kx, "stream_md5": hashlib.md5(streamx).hexdigest(),
                             "bit_exact_vs_reference_encoder": None if refx is None else bool(refx["default"]["stream"] == streamx),
                             "cpu_baseline": None if refx is None else {"value": Kx / refx["default"]["seconds"], "cores": refx["cores"], "kind": "reference", "says": refx["default"]["says"],
                                                                        "pools16": Kx / refx["pools16"]["seconds"]},
                             "reference_command_line": "x265 --input clip.y4m -o out.hevc " + " ".join(clix),
                             "note": "BASELINE.json configs[%d] at its stated size (SURVEY 8d's clip, cfg_id %d), %d frames, the preset as it comes (CRF 28, aq-mode 2, cuTree)" % (cfgIndex, cfg_id, Kx)}
                del framesx, streamx
            except Exception as exc:       # the bench line stands on its own
                line[key] = {"error": repr(exc)}
            finally:
                W, H = 1920, 1080
    # ---- the hot-path kernels on their own (bench_kernels.py): a frame's worth of batched block operations ----
    if not args.no_kernel_workload:
        import bench_kernels
        kargs = bench_kernels.parse_args(["--gpus", str(args.gpus), "--steps", str(max(5, min(20, K))), "--warmup", "3", "--res", args.res] + (["--no-cpu-baseline"] if args.no_cpu_baseline else []))
        kw = bench_kernels.run(kargs)
        if rank == 0 and kw is not None:
            line["kernel_workload"] = {"roofline": dict(kw["roofline"], of="the dominant kernel of this workload (a frame's batched block operations)"),
                                       "frames_per_s": kw["value"], "ms_per_frame": kw["ms_per_step"], "workload": kw["config"]["workload"], "kernels": kw["kernels"],
                                       "parity_sample": kw["parity_sample"], "cpu_baseline": kw.get("cpu_baseline")}
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
