"""Frame-per-GPU with row publication (SURVEY.md section 8e as written; reference analogue: one picture per FrameEncoder, frame k in encode order ->
frame encoder k mod G, source/encoder/encoder.cpp:1872; a finished CTU row is published at the m_reconRowFlag point, framefilter.cpp:654-664, and the
pictures that reference it wait for it at frameencoder.cpp:893-908).

One process per GPU, one encoder object per process opened with shardRank = rank, shardCount = world (include/x265amd_encoder.h).  Every rank is fed
every picture and decides slice types, DPB and reference lists identically; rank r CODES the pictures whose place k in coding order has k % world == r.
The one exchange of the path is the row pump below.

**One publication stream per owner** (round 4).  A consumer of picture k + 1 needs the FIRST rows of picture k + 1's references long before picture k's last
row exists, so rows must not travel in one (picture, row) order over one communicator: every source rank has a process group and a pump thread of its own on
every rank.  The thread of source s walks the pictures s codes, in coding order, rows top to bottom -- the order in which they become final on s -- and
broadcasts each row on s's group; the receivers' threads of that group import what arrives.  Streams of different owners never wait for each other, so G
pictures' rows are in flight at once (what the reference's frame threads do with m_reconRowFlag).  Pictures nobody references (the plain B pictures) are not
sent: `referenced(k)` says so, identically on every rank.  A row travels as ONE buffer -- the three plane ranges and the two map ranges packed behind each
other (5 collectives per row became 1) -- torch.distributed's broadcast is ncclBroadcast = RCCL over xGMI on GPUs and gloo in the CPU test.

The pump is written against callables so that the CPU test can drive it with arrays instead of encoder objects:
    export_row(k, row) -> list of 1-D uint8 tensors (device tensors for the planes, CPU tensors for the maps), blocking until the row is final
    import_row(k, row, tensors) -> None
    shapes(row) -> the byte counts (the geometry is the same on every rank)
    referenced(k) -> bool, optional: whether any later picture may reference picture k (False: its rows stay where they are)."""
import ctypes as C
import threading

import torch
import torch.distributed as dist


def owner_of(coding_index, world):
    return coding_index % world


def _stream(src, export_row, import_row, shapes, n_pictures, ctu_rows, device, rank, world, group, on_row, referenced, errors):
    """the publication stream of source rank `src` on this rank: the pictures src codes, in coding order, rows top to bottom"""
    try:
        bufs = {}
        for k in range(src, n_pictures, world):
            if referenced is not None and not referenced(k):
                continue
            for row in range(ctu_rows):
                sizes = list(shapes(row))
                key = tuple(sizes)
                if world > 1 and key not in bufs:       # one packed buffer per row geometry (first / middle / last row differ by their margins)
                    bufs[key] = torch.empty(sum(sizes), dtype=torch.uint8, device=device)
                if src == rank:
                    tensors = export_row(k, row)
                    if world > 1:
                        buf, at = bufs[key], 0
                        for t, n in zip(tensors, sizes):
                            buf[at:at + n].copy_(t, non_blocking=True)
                            at += n
                        dist.broadcast(buf, src=src, group=group)
                else:
                    buf = bufs[key]
                    dist.broadcast(buf, src=src, group=group)
                    if buf.is_cuda:
                        torch.cuda.current_stream().synchronize()      # the row is complete before any other stream (the importing object's) reads it
                    tensors, at = [], 0
                    for i, n in enumerate(sizes):
                        part = buf[at:at + n]
                        tensors.append(part if i < 3 else part.cpu())   # planes stay on the device; the two small maps are host records
                        at += n
                    import_row(k, row, tensors)
                if on_row:
                    on_row(k, row, src)
    except BaseException as e:          # a stream that dies must not leave the others (and the peers' collectives) hanging without a word
        errors.append((src, e))
        raise


def pump(export_row, import_row, shapes, n_pictures, ctu_rows, device, rank=None, world=None, on_row=None, referenced=None, groups=None):
    """Runs the publication schedule for pictures 0 .. n_pictures - 1 (coding order): one stream per owner (see the module text).  `groups`: a process group per
    source rank (every rank calls pump at the same point, so creating them here is collective too)."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    errors = []
    if world == 1:
        _stream(0, export_row, import_row, shapes, n_pictures, ctu_rows, device, 0, 1, None, on_row, referenced, errors)
        return
    if groups is None:
        groups = [dist.new_group(ranks=list(range(world))) for _ in range(world)]
    threads = [threading.Thread(target=_stream, name="rows-of-rank-%d" % s, daemon=True,
                                args=(s, export_row, import_row, shapes, n_pictures, ctu_rows, device, rank, world, groups[s], on_row, referenced, errors))
               for s in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise RuntimeError("row pump: the stream of rank %d failed: %r" % errors[0])


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

    def _wait(self, call, timeout_s=600.0):
        import time
        t0 = time.monotonic()
        while True:
            rc = call()
            if rc == 0:
                return
            if rc < 0:
                raise RuntimeError(self.lib.x265amd_last_error().decode())
            if time.monotonic() - t0 > timeout_s:
                raise RuntimeError("row pump: the picture did not leave the lookahead within %.0f s" % timeout_s)
            time.sleep(0.0005)          # the picture has not left the lookahead yet

    def referenced(self, k):
        """whether later pictures may reference picture k (x265amd_encoder_is_referenced): known once the picture has left the lookahead"""
        import time
        t0 = time.monotonic()
        while True:
            rc = self.lib.x265amd_encoder_is_referenced(self.enc, k)
            if rc in (0, 1):
                return bool(rc)
            if rc < 0:
                raise RuntimeError(self.lib.x265amd_last_error().decode())
            if time.monotonic() - t0 > 600.0:
                raise RuntimeError("row pump: picture %d did not leave the lookahead" % k)
            time.sleep(0.0005)

    def export_row(self, k, row):
        d = RowExport()
        self._wait(lambda: self.lib.x265amd_encoder_export_row(self.enc, k, row, C.byref(d), 300000))
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
        self._wait(lambda: self.lib.x265amd_encoder_import_row(self.enc, C.byref(d)))

    def _geometry(self, row):
        return self._layout[row]

    def shapes(self, row):
        off, nbytes, ub, mb, uo, mo = self._geometry(row)
        return nbytes + [ub, mb]
