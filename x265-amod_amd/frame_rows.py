"""Frame-per-GPU with row publication (SURVEY.md section 8e as written; reference analogue: one picture per FrameEncoder, frame k in encode order ->
frame encoder k mod G, source/encoder/encoder.cpp:1872; a finished CTU row is published at the m_reconRowFlag point, framefilter.cpp:654-664, and the
pictures that reference it wait for it at frameencoder.cpp:893-908).

One process per GPU, one encoder object per process opened with shardRank = rank, shardCount = world (include/x265amd_encoder.h).  Every rank is fed
every picture and decides slice types, DPB and reference lists identically; rank r CODES the pictures whose place k in coding order has k % world == r.
The one exchange of the path is the row pump below, a thread beside the encode loop: for every picture in coding order and every CTU row, the owner waits
until the row is final (x265amd_encoder_export_row) and broadcasts it -- the filtered samples of the three planes with their margins straight from the
picture in device memory, the row's unit and motion records from the host maps -- and the other ranks hand what they received to
x265amd_encoder_import_row, which copies it into their copy of the picture and opens the gates of the pictures waiting for it (the same counters the
pictures of one object wait on).  torch.distributed's broadcast is ncclBroadcast = RCCL over xGMI on GPUs and gloo in the CPU test; a broadcast is the
natural collective here: one producer, every other rank may reference the row.

The pump is written against two callables so that the CPU test can drive it with arrays instead of encoder objects:
    export_row(k, row) -> list of 1-D uint8 tensors (device tensors for the planes, CPU tensors for the maps), blocking until the row is final
    import_row(k, row, tensors) -> None
`shapes(row)` gives the receiving side the byte counts (the geometry is the same on every rank)."""
import ctypes as C

import torch
import torch.distributed as dist


def owner_of(coding_index, world):
    return coding_index % world


def pump(export_row, import_row, shapes, n_pictures, ctu_rows, device, rank=None, world=None, on_row=None):
    """Runs the publication schedule for pictures 0 .. n_pictures - 1 (coding order).  Rows are published in coding order, top row first: that is the order
    in which any consumer can need them (a picture references only pictures before it in coding order, and its CTU row r reads rows <= r + lag)."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    bufs = {}
    for k in range(n_pictures):
        src = owner_of(k, world)
        for row in range(ctu_rows):
            if src == rank:
                tensors = export_row(k, row)
            else:
                key = tuple(shapes(row))
                if key not in bufs:         # receive buffers per row geometry (first / middle / last row differ by their margins)
                    bufs[key] = [torch.empty(n, dtype=torch.uint8, device=device if i < 3 else "cpu") for i, n in enumerate(key)]
                tensors = bufs[key]
            if world > 1:
                for i, t in enumerate(tensors):
                    # planes travel device to device (RCCL); the two small host maps go through a device tensor on GPUs (NCCL moves device memory only)
                    if t.device.type == "cpu" and device != "cpu":
                        d = t.to(device) if src == rank else torch.empty_like(t, device=device)
                        dist.broadcast(d, src=src)
                        if src != rank:
                            t.copy_(d)
                    else:
                        dist.broadcast(t, src=src)
            if src != rank:
                import_row(k, row, tensors)
            if on_row:
                on_row(k, row, src)


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

    def _wait(self, call):
        import time
        while True:
            rc = call()
            if rc == 0:
                return
            if rc < 0:
                raise RuntimeError(self.lib.x265amd_last_error().decode())
            time.sleep(0.0005)          # the picture has not left the lookahead yet

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
