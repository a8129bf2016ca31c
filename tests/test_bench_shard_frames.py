"""bench.py's `--shard frames` leg (SURVEY.md section 8e as written: picture k in coding order on rank k mod N, finished CTU rows published to the other ranks by
x265-amod_amd/frame_rows.py) run for real: TWO PROCESSES, real encoder objects, real torch.distributed -- on the ONE GPU of the test box, so over the gloo backend (the rows
are staged through the host: RCCL refuses two ranks on one device) with both ranks on cuda:0 and twenty-four resident workgroups each.  (With sixty-four each -- half of the device held by two processes' resident kernels -- four runs in ten stood still: one process's ordinary launches, the in-loop filters', were not scheduled for tens of seconds; with twenty-four, and with none, twelve runs of twelve pass (dbg/shard_dbg.sh).  One process per GPU, the real arrangement, has no second resident kernel beside it.)  The owners' NAL units put back in
coding order must be the single object's stream of the same clip.  The NCCL transport itself needs two GPUs and is not run here; the schedule of collectives is the same."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import hevc_testlib as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FRAMES = 10


@pytest.mark.gpu
def test_bench_shard_frames_two_processes_one_gpu(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "stream")
    env = dict(os.environ, X265AMD_QUEUES="24", X265AMD_BENCH_STREAM_OUT=out, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(FRAMES), "--warmup", "0", "--shard", "frames", "--backend", "gloo", "--one-gpu",
           "--no-kernel-workload", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["steps"] == FRAMES
    parts = []
    for rank in range(2):
        data = open("%s.%d" % (out, rank), "rb").read()
        marks = json.load(open("%s.%d.marks" % (out, rank)))
        assert len(marks) == FRAMES                         # every object returns every picture, in coding order: the ones it does not code come without NAL units
        chunks = [data[a:b] for a, b in zip(marks, marks[1:] + [len(data)])]
        assert all((len(c) > 0) == (k % 2 == rank) for k, c in enumerate(chunks)), [len(c) for c in chunks]
        parts.append((data[:marks[0]], chunks))
    together = parts[0][0] + b"".join(parts[k % 2][1][k] for k in range(FRAMES))
    # the same clip through ONE object in this process (the ranks have left the GPU)
    sys.path.insert(0, ROOT)
    import bench
    frames = T.survey_clip(1920, 1080, 8, 2, 0, FRAMES)
    stream, coded = T.encoder_run(T.load_hip(8), frames, 1920, 1080, **bench.ENC_CFG)
    assert len(coded) == FRAMES
    assert not T.stream_diff(np.frombuffer(together, dtype=np.uint8), stream), T.stream_diff(np.frombuffer(together, dtype=np.uint8), stream)
