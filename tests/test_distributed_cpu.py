"""The N>1 path on CPU: world_size 2 over gloo.  Each rank fills the pixels of its round-robin
tiles (the oracle stands in for the device renderer here -- this test is about the partition and
the reduce, which are the only multi-GPU logic the path has) and the sum-reduce on rank 0 must be
bit-identical to the undivided image."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

import oracle_lib as ol  # noqa: E402
from simple_spectral_amd import dist as sdist  # noqa: E402

W, H, SPP = 44, 27, 3


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = ol.Oracle("cornell-srgb", texture="test-img.png")
    full = o.render(W, H, SPP, seed=5, nthreads=1)
    mine = np.where(sdist.tile_owner_mask(W, H, rank, world, skew=1)[..., None], full, np.float32(0))   # (bench.py's partition: tile rows rotated)
    t = torch.from_numpy(np.ascontiguousarray(mine))
    sdist.reduce_framebuffer(t, dst=0)
    if rank == 0:
        q.put((t.numpy().copy(), full))
    dist.barrier()
    dist.destroy_process_group()


def test_tile_partition_is_a_partition():
    for skew in (0, 1, 3):
        for world in (1, 2, 3, 4, 8):
            masks = [sdist.tile_owner_mask(100, 61, r, world, skew) for r in range(world)]
            assert np.array_equal(np.sum(masks, axis=0), np.ones((61, 100)))
            # 8x8 granularity
            m = masks[0]
            assert m[:8, :8].all() or not m[:8, :8].any()
    # 64 tiles per row and 8 ranks: the plain list gives rank 0 whole tile columns, the rotated one a share of every column
    plain, rotated = sdist.tile_owner_mask(512, 512, 0, 8, 0), sdist.tile_owner_mask(512, 512, 0, 8, 1)
    assert set(plain.sum(axis=0)) == {0, 512} and set(rotated.sum(axis=0)) == {64}
    assert plain.sum() == rotated.sum() == 512 * 512 // 8


def test_two_rank_gloo_reduce_reassembles_the_image():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, full = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert np.array_equal(got.view(np.uint32), full.view(np.uint32))
