"""N>1 paths on CPU (gloo): closed GOPs per rank (x265-amod_amd/gop_shard.py) and pictures per rank with their CTU rows broadcast (x265-amod_amd/frame_rows.py: what bench.py
runs on RCCL)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


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


# ---- frame-per-GPU with row publication (x265-amod_amd/frame_rows.py): the schedule, with arrays standing in for the encoder objects ----
ROWS, PICS = 5, 7
if os.environ.get("X265AMD_TEST_PICS"):
    PICS = int(os.environ["X265AMD_TEST_PICS"])


def _row_bytes(k, row, i, n):
    return torch.from_numpy(np.random.default_rng(k * 1000 + row * 10 + i).integers(0, 256, n, dtype=np.uint8))


def _shapes(row):
    # first / last rows carry the top / bottom margin: three row geometries, as in the encoder
    return [4096 + (512 if row in (0, ROWS - 1) else 0), 1024, 1024, 160, 96]


class _FakeEncoder:
    """holds the rows it has (its own and the imported ones); a coding thread makes this rank's rows final one after the other: a row of picture k can only be coded
    when rows <= row + 1 of picture k - 1 are there.  The LAST row of a picture is held back until row 0 of the NEXT picture -- coded by the other rank, from this
    picture's first rows -- has arrived: with one ordered stream of (picture, row) pairs that never happens (the pump would be waiting for this very row); the pump
    must carry whichever owner's rows are final.  Pictures in `late` are not known to this object at first: import_row says False three times (the pump keeps the row)."""

    def __init__(self, rank, world, skipped=(), pics=None, late=()):
        import threading
        self.rank, self.world, self.have, self.order, self.skipped = rank, world, {}, [], set(skipped)
        self.pics = PICS if pics is None else pics
        self.cv = threading.Condition()
        self.final = set()
        self.late, self.refused = set(late), 0
        self.error = None
        self.coder = threading.Thread(target=self._code, daemon=True)
        self.coder.start()

    def _wait_for(self, key, what):
        with self.cv:
            if not self.cv.wait_for(lambda: key in self.have, timeout=60):
                raise AssertionError(what)

    def _code(self):
        try:
            for k in range(self.rank, self.pics, self.world):
                for row in range(ROWS):
                    ref = k - 1
                    while ref in self.skipped:          # a picture that does not travel is nobody's reference
                        ref -= 1
                    if ref >= 0 and k not in self.skipped and self.world > 1:
                        for r in range(min(row + 2, ROWS)):
                            self._wait_for((ref, r), "row %d of picture %d never arrived (needed by row %d of picture %d)" % (r, ref, row, k))
                    if row == ROWS - 1 and k + 1 < self.pics and k + 1 not in self.skipped and self.world > 1:
                        self._wait_for((k + 1, 0), "row 0 of picture %d did not travel before the last row of picture %d" % (k + 1, k))
                    with self.cv:
                        self.have[(k, row)] = [_row_bytes(k, row, i, n) for i, n in enumerate(_shapes(row))]
                        self.final.add((k, row))
                        self.cv.notify_all()
        except BaseException as e:      # noqa: B902
            self.error = e

    def export_row(self, k, row):
        assert k % self.world == self.rank
        if self.error is not None:
            raise self.error
        with self.cv:
            if (k, row) not in self.final:
                return None
            self.order.append((k, row, "export"))
            return [x.clone() for x in self.have[(k, row)]]

    def import_row(self, k, row, tensors):
        assert k % self.world != self.rank and (k, row) not in self.have
        with self.cv:
            if k in self.late:
                self.refused += 1
                if self.refused >= 3:
                    self.late.discard(k)       # known from the next look on
                return False
            self.have[(k, row)] = [x.clone() for x in tensors]
            self.order.append((k, row, "import"))
            self.cv.notify_all()
        return True


def _rows_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as g
    fr = g.load_package().frame_rows
    skipped = {3}                           # a picture nobody references: its rows stay with its owner
    pick = max if os.environ.get("X265AMD_TEST_LATE_LAST") else min      # the LAST picture this rank receives: its rows are still pending when every sender is done
    late = {pick(k for k in range(4, PICS) if k % world != rank and k not in skipped)} if rank == 1 else ()
    enc = _FakeEncoder(rank, world, skipped, late=late)
    answers = {}

    def referenced(k):                      # the first question about a picture is answered "not known yet"
        answers[k] = answers.get(k, 0) + 1
        return None if answers[k] == 1 else k not in skipped
    fr.pump(enc.export_row, enc.import_row, _shapes, PICS, ROWS, "cpu", referenced=referenced, idle_timeout_s=60.0)
    sent = [k for k in range(PICS) if k not in skipped]
    ok = set(enc.have) == {(k, r) for k in sent for r in range(ROWS)} | {(k, r) for k in skipped if k % world == rank for r in range(ROWS)}
    for (k, row), t in enc.have.items():
        ok &= all(torch.equal(a, _row_bytes(k, row, i, n)) for i, (a, n) in enumerate(zip(t, _shapes(row))))
    # per owner the rows arrive in coding order, top row first; between owners they interleave: row 0 of picture k + 1 is there before the last row of picture k
    for s in range(world):
        mine = [(k, r) for k, r, _ in enc.order if fr.owner_of(k, world) == s]
        ok &= mine == [(k, r) for k in sent if fr.owner_of(k, world) == s for r in range(ROWS)]
    ok &= all((what == "export") == (fr.owner_of(k, world) == rank) for k, r, what in enc.order)
    ok &= enc.error is None and (enc.refused == 3 if late else enc.refused == 0)
    dist.barrier()
    out[rank] = bool(ok)
    dist.destroy_process_group()


def _spawn_rows(world, worker=None):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(worker or _rows_worker, args=(world, port, out), nprocs=world, join=True)
    return out


def test_row_publication_schedule_gloo_world2():
    """x265-amod_amd/frame_rows.py: picture k in coding order is coded by rank k % 2; each finished CTU row is broadcast by its owner and imported by the other rank
    before that rank codes the rows that reference it (the fake encoders' coding threads stand still otherwise: the pump would time out)"""
    out = _spawn_rows(2)
    assert out[0] and out[1]


def test_row_publication_schedule_gloo_world3():
    """the same schedule with three owners: picture k's rows come from rank k % 3 and reach both others"""
    out = _spawn_rows(3)
    assert out[0] and out[1] and out[2]


def test_row_publication_last_picture_late_on_a_receiver(monkeypatch):
    """the last referenced picture is not known to rank 1 when its rows arrive -- they stay pending while every sender has nothing left to send.  The pump
    must keep ALL ranks in its slots until rank 1 has imported them (a rank that leaves alone hangs the others' next collective), and every row must arrive"""
    monkeypatch.setenv("X265AMD_TEST_LATE_LAST", "1")
    out = _spawn_rows(2)
    assert out[0] and out[1]


def test_row_publication_schedule_gloo_world8(monkeypatch):
    """the schedule of an eight-GPU node: picture k's rows come from rank k % 8 and reach the seven others (the same fake encoders; 16 pictures so that every rank owns two)"""
    monkeypatch.setenv("X265AMD_TEST_PICS", "16")           # (the spawned ranks import this module afresh)
    out = _spawn_rows(8)
    assert all(out[r] for r in range(8))


def _failing_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as g
    fr = g.load_package().frame_rows
    enc = _FakeEncoder(rank, world)

    def export_row(k, row):
        if rank == 1 and (k, row) == (1, 2):
            raise ValueError("the picture failed")
        return enc.export_row(k, row)
    try:
        fr.pump(export_row, enc.import_row, _shapes, PICS, ROWS, "cpu", idle_timeout_s=60.0)
        out[rank] = "returned"
    except RuntimeError as e:
        out[rank] = str(e)
    dist.destroy_process_group()


def test_row_publication_failure_reaches_every_rank():
    """a rank whose local step fails says so in its next header: every rank raises instead of hanging in a collective"""
    out = _spawn_rows(2, _failing_worker)
    assert "rank 1 failed" in out[1] and "failed" in out[0] and "returned" not in (out[0], out[1])


def test_row_publication_single_rank_is_a_no_op():
    import __graft_entry__ as g
    fr = g.load_package().frame_rows
    enc = _FakeEncoder(0, 1, pics=3)
    fr.pump(enc.export_row, enc.import_row, _shapes, 3, ROWS, "cpu", rank=0, world=1)
    assert len(enc.have) == 3 * ROWS and all(w == "export" for _, _, w in enc.order)
    assert [(k, r) for k, r, _ in enc.order] == [(k, r) for k in range(3) for r in range(ROWS)]
