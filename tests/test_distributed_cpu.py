"""N>1 path on CPU: world_size-2 gloo run of the frame sharding + publish (all_gather) schedule that bench.py uses on
RCCL (x265-amod_amd/frame_shard.py)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _picture(idx, n):
    rng = np.random.default_rng(idx)
    return torch.from_numpy(rng.integers(0, 256, n, dtype=np.uint8))


def _worker(rank, world, port, steps, n, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as g
    fs = g.load_package().frame_shard
    ring = fs.ReferenceRing(depth=3 * world)
    buf = None
    ok = True
    for step in range(steps):
        mine = fs.frames_of_step(step, world)[rank]
        assert fs.frame_owner(mine, world) == rank
        buf = fs.publish_step(_picture(mine, n), step, ring, buf)
        for idx in fs.frames_of_step(step, world):
            ok &= bool(torch.equal(ring.get(idx), _picture(idx, n)))
        # pictures older than the ring depth are dropped, newer ones are all present
        ok &= all(ring.has(i) for i in range(max(0, (step + 1) * world - 3 * world), (step + 1) * world))
    dist.barrier()
    out[rank] = ok
    dist.destroy_process_group()


def test_frame_shard_gloo_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, 5, 4096, out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}


def test_owner_schedule():
    import __graft_entry__ as g
    fs = g.load_package().frame_shard
    assert [fs.frame_owner(k, 8) for k in range(10)] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1]
    assert fs.frames_of_step(3, 4) == [12, 13, 14, 15]


def _gop_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as g
    gs = g.load_package().gop_shard
    coded = []

    def encode_gop(first, end):          # stands in for an encoder object started at `first`: the bytes name the pictures it was given
        coded.append((first, end))
        return b"".join(b"<%d>" % i for i in range(first, end))

    stream = gs.encode_sharded(23, 5, encode_gop, lambda: b"HDR")
    want_mine = [r for k, r in enumerate(gs.gop_ranges(23, 5)) if k % world == rank]
    ok = coded == want_mine and (stream == b"HDR" + b"".join(b"<%d>" % i for i in range(23)) if rank == 0 else stream is None)
    dist.barrier()
    out[rank] = ok
    dist.destroy_process_group()


def test_gop_sharding_schedule_and_gather_world2():
    """the encoder object's multi-GPU path (x265-amod_amd/gop_shard.py): closed GOPs round-robin over the ranks, coded GOPs gathered on rank 0 in order"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_gop_worker, args=(2, port, out), nprocs=2, join=True)
    assert out[0] and out[1]
