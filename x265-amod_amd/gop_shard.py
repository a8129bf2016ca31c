"""Multi-GPU encoding with the encoder object (include/x265amd_encoder.h): closed GOPs are the independent units.

The reference shards an encode over frame threads that exchange reconstructed rows.  With closed GOPs (--no-open-gop: every
keyframe is an IDR picture) nothing crosses a GOP boundary -- reference pictures, motion fields, the SAO rate state the reference carries from picture
to picture (an IDR picture rewrites it before any later picture reads it) -- so rank r of N takes GOPs r, r + N, ... and codes each with an encoder
object started at that GOP's first picture (`x265amd_param.firstFrame`).  There is no data-path collective: the only communication is the gather of the
coded GOPs on rank 0, which writes them in order behind the stream headers.  The result is the single-encoder byte stream
(tests/test_encoder_api.py::test_closed_gops_encode_independently checks the independence on one GPU, tests/test_distributed_cpu.py the schedule and the
gather with two gloo ranks)."""
import torch.distributed as dist


def gop_ranges(num_frames, keyint):
    """[first, last+1) display-order picture ranges of the closed GOPs of a fixed --keyint / --min-keyint = keyint encode"""
    return [(s, min(s + keyint, num_frames)) for s in range(0, num_frames, keyint)]


def gops_of_rank(num_frames, keyint, rank, world):
    return [(g, r) for g, r in enumerate(gop_ranges(num_frames, keyint)) if g % world == rank]


def encode_sharded(num_frames, keyint, encode_gop, headers):
    """encode_gop(first, end) -> bytes of the GOP's slice NAL units (an encoder object with firstFrame = first, fed pictures first .. end - 1);
    headers() -> the VPS / SPS / PPS bytes (rank 0 only).  Returns the whole stream on rank 0, None elsewhere.  Works without an initialised
    process group (one rank)."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
    mine = [(g, encode_gop(first, end)) for g, (first, end) in gops_of_rank(num_frames, keyint, rank, world)]
    if world == 1:
        gathered = [mine]
    else:
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(mine, gathered, dst=0)
    if rank != 0:
        return None
    parts = dict(p for per_rank in gathered for p in per_rank)
    return b"".join([headers()] + [parts[g] for g in range(len(gop_ranges(num_frames, keyint)))])
