"""Frame-per-GPU with row publication (SURVEY.md section 8e as written; reference analogue: one picture per FrameEncoder, frame k in encode order ->
frame encoder k mod G, source/encoder/encoder.cpp:1872; a finished CTU row is published at the m_reconRowFlag point, framefilter.cpp:654-664, and the
pictures that reference it wait for it at frameencoder.cpp:893-908).

One process per GPU, one encoder object per process opened with shardRank = rank, shardCount = world (include/x265amd_encoder.h).  Every rank is fed
every picture and decides slice types, DPB and reference lists identically; rank r CODES the pictures whose place k in coding order has k % world == r.
The one exchange of the path is the row pump below.

**One thread, one communicator, time slots** (round 5; SURVEY 8e: "one all-gather per frame-time slot where each rank contributes the rows it finished").
A consumer of picture k + 1 needs the FIRST rows of picture k + 1's references long before picture k's last row exists, so rows must not travel in one fixed
(picture, row) order.  Round 4 gave every owner a communicator and a thread of its own; collectives of several communicators issued from several threads in
different orders on different ranks can deadlock under NCCL / RCCL (every rank's stream may hold the other's collective behind its own), and the threads shared
the process's current stream.  Now every rank runs ONE pump thread on a stream of its own and all ranks issue the SAME sequence of collectives:

    slot:  every rank LOOKS (without waiting) which of its own next rows are final        -> header (picture, first row, count, state)
           the headers are exchanged (one all-reduce of a world x 4 table: works on NCCL and on gloo, device or host tensors)
           for every rank s with count > 0, in rank order: ONE broadcast of those rows, packed behind each other, from s
           the receivers import what arrived; a row whose picture the local object does not know yet (its lookahead has not handed it over) is kept and retried

Nobody waits for a row inside a collective, so the streams of different owners never hold each other up: G pictures' rows are in flight at once (what the reference's
frame threads do with m_reconRowFlag).  Pictures nobody references (the plain B pictures) are not sent (`referenced(k)`).  A rank whose local step failed says so in
its next header and every rank raises: nobody is left hanging in a collective.  torch.distributed's broadcast is ncclBroadcast = RCCL over xGMI on GPUs and gloo in the
CPU tests.

The pump is written against callables so that the CPU test can drive it with arrays instead of encoder objects:
    export_row(k, row) -> list of 1-D uint8 tensors (device tensors for the planes, CPU tensors for the maps) or None when the row is not final yet (never waits)
    import_row(k, row, tensors) -> None / True when done, False when the picture is not known here yet (the pump keeps the row and asks again)
    shapes(row) -> the byte counts (the geometry is the same on every rank)
    referenced(k) -> True / False, or None while not known yet; optional: whether any later picture may reference picture k (False: its rows stay where they are)."""
import ctypes as C
import time

import torch
import torch.distributed as dist

ROWS_PER_SLOT = 4           # at most this many rows of one owner travel in one broadcast
STATE_MORE, STATE_DONE, STATE_FAILED = 0, 1, 2


def owner_of(coding_index, world):
    return coding_index % world


class _Cursor:
    """the rows this rank has still to publish: its pictures in coding order, rows top to bottom, the unreferenced pictures left out"""

    def __init__(self, rank, world, n_pictures, ctu_rows, referenced):
        self.k, self.row, self.world, self.n, self.rows, self.referenced, self.known = rank, 0, world, n_pictures, ctu_rows, referenced, False

    def done(self):
        return self.k >= self.n

    def settle(self):
        """skips pictures that do not travel; False while the answer for the current picture is not known yet"""
        while self.k < self.n and not self.known:
            r = True if self.referenced is None else self.referenced(self.k)
            if r is None:
                return False
            if r:
                self.known = True
            else:
                self.k += self.world
        return True

    def advance(self):
        self.row += 1
        if self.row == self.rows:
            self.row, self.k, self.known = 0, self.k + self.world, False


def _pump_world1(export_row, n_pictures, ctu_rows, on_row, referenced):
    cur = _Cursor(0, 1, n_pictures, ctu_rows, referenced)
    t_idle = time.monotonic()
    while not cur.done():
        if cur.settle() and not cur.done():
            t = export_row(cur.k, cur.row)
            if t is not None:
                if on_row:
                    on_row(cur.k, cur.row, 0)
                cur.advance()
                t_idle = time.monotonic()
                continue
        if time.monotonic() - t_idle > 600.0:
            raise RuntimeError("row pump: nothing became final for 600 s")
        time.sleep(0.0002)


def pump(export_row, import_row, shapes, n_pictures, ctu_rows, device, rank=None, world=None, on_row=None, referenced=None, group=None, idle_timeout_s=600.0):
    """Runs the publication schedule for pictures 0 .. n_pictures - 1 (coding order) on the calling thread (see the module text).  `group`: the process group to use
    (None: the default one).  On a GPU the collectives, the packing copies and the waits run on a stream of this call's own."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
        _pump_world1(export_row, n_pictures, ctu_rows, on_row, referenced)
        return
    dev = torch.device(device)
    stream = None
    if dev.type == "cuda":
        torch.cuda.set_device(dev)                      # a new thread starts on device 0: the collectives and the waits below belong to THIS rank's device
        stream = torch.cuda.Stream(dev)
    ctx = torch.cuda.stream(stream) if stream is not None else _NullCtx()
    with ctx:
        _slots(export_row, import_row, shapes, n_pictures, ctu_rows, dev, stream, rank, world, on_row, referenced, group, idle_timeout_s)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _slots(export_row, import_row, shapes, n_pictures, ctu_rows, dev, stream, rank, world, on_row, referenced, group, idle_timeout_s):
    cur = _Cursor(rank, world, n_pictures, ctu_rows, referenced)
    state = [STATE_MORE] * world
    pending = []                    # (k, row, tensors) received but not importable yet
    failure = None
    cur_first = (0, 0)
    t_moved = time.monotonic()

    def local(step):
        """a local step must not leave the peers hanging in the next collective: its failure travels in the next header"""
        nonlocal failure
        if failure is None:
            try:
                return step()
            except BaseException as e:      # noqa: B902
                failure = e
        return None

    def try_imports():
        moved = False
        keep = []
        unknown = set()             # a picture refused once is left alone for the rest of this pass: its rows go in in order
        for k, row, tensors in pending:
            if k in unknown:
                keep.append((k, row, tensors))
            elif failure is None and local(lambda: import_row(k, row, tensors)) is not False and failure is None:
                moved = True
                if on_row:
                    on_row(k, row, owner_of(k, world))
            else:
                unknown.add(k)
                keep.append((k, row, tensors))
        pending[:] = keep
        return moved

    while True:
        # ---- what this rank has ready (a look, never a wait) ----
        mine = []
        if failure is None and not cur.done() and local(cur.settle):
            first = (cur.k, cur.row)
            while not cur.done() and cur.k == first[0] and len(mine) < ROWS_PER_SLOT:
                t = local(lambda: export_row(cur.k, cur.row))
                if t is None:
                    break
                mine.append(t)
                if on_row:
                    on_row(cur.k, cur.row, rank)
                cur.advance()
            if mine:
                cur_first = first
        # DONE means: nothing left to send AND nothing received that is still waiting for its picture -- the end of the pump is decided from the exchanged table alone
        # (below), so a rank that still holds rows keeps every rank in the loop with it instead of walking into the next collective alone
        my_state = STATE_FAILED if failure is not None else (STATE_DONE if cur.done() and not mine and not pending else STATE_MORE)
        hdr = torch.zeros(world, 4, dtype=torch.int64)
        hdr[rank] = torch.tensor([cur_first[0] if mine else 0, cur_first[1] if mine else 0, len(mine), my_state], dtype=torch.int64)
        hdr = hdr.to(dev)
        dist.all_reduce(hdr, group=group)               # every rank filled its own line only: the sum is the table
        table = hdr.cpu().tolist()
        if any(line[3] == STATE_FAILED for line in table):
            who = [s for s, line in enumerate(table) if line[3] == STATE_FAILED]
            if failure is not None:
                raise RuntimeError("row pump: rank %d failed: %r" % (rank, failure)) from failure
            raise RuntimeError("row pump: rank(s) %s failed; this rank stops with them" % who)
        moved = False
        for s in range(world):
            k, row0, n, st = table[s]
            state[s] = st
            if not n:
                continue
            moved = True
            sizes = [list(shapes(row0 + j)) for j in range(n)]
            total = sum(sum(z) for z in sizes)
            buf = torch.empty(total, dtype=torch.uint8, device=dev)
            if s == rank:
                at = 0
                for tensors, z in zip(mine, sizes):
                    for t, m in zip(tensors, z):
                        buf[at:at + m].copy_(t, non_blocking=True)
                        at += m
            dist.broadcast(buf, src=s, group=group)
            if s != rank:
                if stream is not None:
                    stream.synchronize()            # the rows are complete before any other stream (the importing object's) reads them
                at = 0
                for j, z in enumerate(sizes):
                    tensors = []
                    for i, m in enumerate(z):
                        part = buf[at:at + m]
                        tensors.append(part if i < 3 else part.cpu())       # planes stay on the device; the two small maps are host records
                        at += m
                    pending.append((k, row0 + j, tensors))
            elif stream is not None:
                stream.synchronize()                # the packing copies have read the picture
        moved |= try_imports()
        if all(line[3] == STATE_DONE and line[2] == 0 for line in table):
            return                                  # every rank said so in THIS slot and nothing travelled in it: nothing new can be pending anywhere
        if moved:
            t_moved = time.monotonic()
        else:
            if time.monotonic() - t_moved > idle_timeout_s:
                failure = failure or RuntimeError("row pump: nothing moved for %.0f s (%d rows kept for pictures this rank does not know yet)" % (idle_timeout_s, len(pending)))
                continue                           # the next header carries it
            time.sleep(0.0002)


class RowExport(C.Structure):       # x265amd_row_export (include/x265amd_encoder.h)
    _fields_ = [("coding_index", C.c_uint64), ("ctu_row", C.c_int32), ("reserved", C.c_int32), ("src", C.c_void_p * 3), ("plane_offset", C.c_uint64 * 3), ("plane_bytes", C.c_uint64 * 3),
                ("units", C.c_void_p), ("units_bytes", C.c_uint64), ("motion", C.c_void_p), ("motion_bytes", C.c_uint64), ("map_offset_units", C.c_uint64), ("map_offset_motion", C.c_uint64)]


class _DevView:
    """a range of device memory as a tensor (no copy): torch takes it through __cuda_array_interface__"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class EncoderRows:
    """export_row / import_row / shapes of one encoder object (lib: the loaded libx265amd_main*.so, enc: its handle)"""

    def __init__(self, lib, enc, device):
        self.lib, self.enc, self.device = lib, enc, device
        lib.x265amd_encoder_export_row.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(RowExport), C.c_int]
        lib.x265amd_encoder_import_row.argtypes = [C.c_void_p, C.POINTER(RowExport)]
        lib.x265amd_encoder_ctu_rows.argtypes = [C.c_void_p]
        lib.x265amd_encoder_is_referenced.argtypes = [C.c_void_p, C.c_uint64]
        lib.x265amd_last_error.restype = C.c_char_p
        lib.x265amd_encoder_row_geometry.argtypes = [C.c_void_p, C.c_int, C.POINTER(RowExport)]
        self.rows = lib.x265amd_encoder_ctu_rows(enc)
        self._layout = {}
        for row in range(self.rows):        # the ranges of every row: the same on every rank of the set
            d = RowExport()
            if lib.x265amd_encoder_row_geometry(enc, row, C.byref(d)) != 0:
                raise RuntimeError(lib.x265amd_last_error().decode())
            self._layout[row] = ([int(d.plane_offset[i]) for i in range(3)], [int(d.plane_bytes[i]) for i in range(3)], int(d.units_bytes), int(d.motion_bytes),
                                 int(d.map_offset_units), int(d.map_offset_motion))

    def referenced(self, k):
        """whether later pictures may reference picture k (x265amd_encoder_is_referenced); None while the picture has not left the lookahead"""
        rc = self.lib.x265amd_encoder_is_referenced(self.enc, k)
        if rc in (0, 1):
            return bool(rc)
        if rc < 0:
            raise RuntimeError(self.lib.x265amd_last_error().decode())
        return None

    def export_row(self, k, row):
        """the row's tensors, or None while it is not final (x265amd_encoder_export_row with timeout 0: a look, not a wait)"""
        d = RowExport()
        rc = self.lib.x265amd_encoder_export_row(self.enc, k, row, C.byref(d), 0)
        if rc < 0:
            raise RuntimeError(self.lib.x265amd_last_error().decode())
        if rc != 0:
            return None
        planes = [torch.as_tensor(_DevView(int(d.src[i]), int(d.plane_bytes[i])), device=self.device) for i in range(3)]
        units = torch.frombuffer((C.c_ubyte * d.units_bytes).from_address(d.units), dtype=torch.uint8)
        motion = torch.frombuffer((C.c_ubyte * d.motion_bytes).from_address(d.motion), dtype=torch.uint8)
        return planes + [units, motion]

    def import_row(self, k, row, tensors):
        off, nbytes, ub, mb, uo, mo = self._geometry(row)
        d = RowExport()
        d.coding_index, d.ctu_row = k, row
        for i in range(3):
            d.src[i] = tensors[i].data_ptr(); d.plane_offset[i] = off[i]; d.plane_bytes[i] = nbytes[i]
        d.units, d.units_bytes, d.map_offset_units = tensors[3].data_ptr(), ub, uo
        d.motion, d.motion_bytes, d.map_offset_motion = tensors[4].data_ptr(), mb, mo
        rc = self.lib.x265amd_encoder_import_row(self.enc, C.byref(d))
        if rc < 0:
            raise RuntimeError(self.lib.x265amd_last_error().decode())
        return rc == 0          # False: the picture has not left this object's lookahead yet -- the pump keeps the row and asks again

    def _geometry(self, row):
        return self._layout[row]

    def shapes(self, row):
        off, nbytes, ub, mb, uo, mo = self._geometry(row)
        return nbytes + [ub, mb]
