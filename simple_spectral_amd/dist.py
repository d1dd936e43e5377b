"""Multi-GPU plumbing: one process per GPU, tiles dealt round-robin, one reduce at the end.

The reference scales with N threads pulling 8x8 tiles off one list (src/renderer.cpp:340-379,
396-409); across GPUs the same tile list is dealt statically: rank r renders tiles t with
t % world == r (interleaving keeps the per-rank cost even although cost per pixel is not), into a
zero-initialised full-size float4 XYZA buffer.  Seeds depend only on (seed, pixel, k), so the
union is bit-identical to a single-device render, and since x + 0 == x exactly the combine is one
sum-reduce (RCCL over xGMI on the GPU box; gloo in the CPU tests).
"""
import numpy as np

TILE = 8  # TILE_SIZE, src/stdafx.hpp:50


def tile_grid(width, height):
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def tile_owner_mask(width, height, rank, world, skew=0):
    """Boolean [H, W] mask of the pixels whose 8x8 tile belongs to `rank`: tile t' of the shared-out list with t' % world == rank,
    the list being the row-major one with tile row ty rotated by ty * skew columns (ssx_render_params.tile_skew; 0: plain)."""
    tx, _ = tile_grid(width, height)
    jj, ii = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
    tile = (jj // TILE) * tx + (ii // TILE + (jj // TILE) * skew) % tx
    return (tile % world) == rank


def reduce_framebuffer(tensor, dst=0):
    """Sum-reduce the per-rank XYZA buffers onto `dst` (torch.distributed must be initialised).
    Message size = 16 bytes * W * H: 4 MiB at 512^2, 16 MiB at 1024^2, 64 MiB at 2048^2."""
    import torch.distributed as dist

    dist.reduce(tensor, dst=dst, op=dist.ReduceOp.SUM)
    return tensor
