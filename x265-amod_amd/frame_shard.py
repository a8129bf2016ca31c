"""Whole-picture exchange between GPUs for the KERNEL workload only (bench_kernels.py --gpus N): superseded for the encoder by frame_rows.py (rows published as they
become final, SURVEY.md section 8e as written) -- kept because bench_kernels.py and one gloo test still time / exercise the all_gather form.

Frame-parallel sharding across GPUs (reference analogue: frame threads, one frame per
FrameEncoder: source/encoder/encoder.cpp:306-328, frameencoder.cpp:285-302).

Frame k in encode order is owned by rank k % world (the reference's m_curEncoder round-robin, encoder.cpp:1872).  The one
exchange step of the path: when an owner has finished a picture that later pictures reference, it publishes the padded
reconstruction to every rank (the point where the reference sets Frame::m_reconRowFlag, framefilter.cpp:654-664).
Implemented as one all_gather per step over torch.distributed (backend nccl = RCCL on GPUs, gloo in the CPU tests):
after step s every rank holds the pictures s*world .. s*world+world-1.
"""
import torch
import torch.distributed as dist


def frame_owner(frame_idx, world):
    return frame_idx % world


def frames_of_step(step, world):
    """encode-order indices of the pictures processed (one per rank) in `step`"""
    return [step * world + r for r in range(world)]


class ReferenceRing:
    """per-rank store of the most recent `depth` published pictures, keyed by encode-order index"""

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


def publish_step(local_plane, step, ring, gather_buf=None):
    """all ranks call this once per step with the picture they just finished (uint8 tensor, same numel everywhere).
    Afterwards `ring` holds the pictures of every rank for this step.  Returns the gather buffer for reuse."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
        ring.put(step, local_plane)
        return None
    if gather_buf is None:
        gather_buf = torch.empty(world * local_plane.numel(), dtype=local_plane.dtype, device=local_plane.device)
    dist.all_gather_into_tensor(gather_buf, local_plane.reshape(-1))
    n = local_plane.numel()
    for r, idx in enumerate(frames_of_step(step, world)):
        ring.put(idx, gather_buf[r * n:(r + 1) * n].clone() if r != rank else local_plane)
    return gather_buf
