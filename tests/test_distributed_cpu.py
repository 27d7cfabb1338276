"""N>1 path on CPU: the screen-tile partition, the gather layout and the de-tiling, with torch.distributed gloo, world_size 2.

No GPU here, so each rank fills its tile-major buffer from a frame rendered by the ORACLE (checker) with the same host-side
partition function the kernel uses (tile k -> rank k % world, 8x8 pixels, tile-major [tiles_per_rank][64]); the ranks gather the
buffers to rank 0 exactly as bench.py does over RCCL, and the de-tiled result must be the original frame."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _tile_major(frame, world, rank, rrt):
    h, w = frame.shape
    tx, ty = (w + 7) // 8, (h + 7) // 8
    tpr = rrt.tiles_per_rank(w, h, world)
    padded = np.zeros((ty * 8, tx * 8), np.uint32); padded[:h, :w] = frame
    tiles = padded.reshape(ty, 8, tx, 8).transpose(0, 2, 1, 3).reshape(tx * ty, 64)
    mine = np.zeros((tpr, 64), np.uint32)
    own = tiles[rank::world]
    mine[:len(own)] = own
    return mine


def _worker(rank, world, port, w, h, frame_path, out_dir):
    sys.path.insert(0, ROOT)
    import importlib
    rrt = importlib.import_module("rust-ray-tracer_amd")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frame = np.load(frame_path)
    mine = torch.from_numpy(_tile_major(frame, world, rank, rrt).view(np.int32).copy()).reshape(-1)
    gathered = torch.empty(world * mine.numel(), dtype=torch.int32)
    chunks = [gathered[i * mine.numel():(i + 1) * mine.numel()] for i in range(world)] if rank == 0 else None
    dist.gather(mine, chunks, dst=0)                       # as bench.py: final gather to rank 0, which de-tiles
    if rank == 0:
        fb = rrt.detile_host(gathered.numpy().view(np.uint32), w, h, world)
        np.save(os.path.join(out_dir, "fb_0.npy"), fb)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("w,h", [(64, 48), (75, 37)])
def test_tile_partition_allgather_gloo_world2(rrt, teapot_oracle, tmp_path, w, h):
    frame, _ = teapot_oracle.render(w, h)
    np.save(tmp_path / "frame.npy", frame)
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, w, h, str(tmp_path / "frame.npy"), str(tmp_path)), nprocs=world, join=True)
    assert np.array_equal(np.load(tmp_path / "fb_0.npy"), frame)


def test_partition_is_balanced_and_complete(rrt):
    for (w, h, world) in [(1920, 1080, 8), (3840, 2160, 8), (75, 37, 3), (8, 8, 4)]:
        owner = rrt.tile_owner_map(w, h, world)
        counts = np.bincount(owner.ravel(), minlength=world)
        assert counts.max() - counts.min() <= 1 and counts.max() == rrt.tiles_per_rank(w, h, world)
