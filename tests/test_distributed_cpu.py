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


def _pipelined_worker(rank, world, port, w, h, frame_path, out_dir, steps, depth):
    """bench.py's N > 1 step with frames in flight: per-slot tile and gather buffers, asynchronous gathers, a slot is reused only after its
    previous gather has completed and its gather buffer has been de-tiled.  Frame i is the base frame + i, so a mixed-up slot shows."""
    sys.path.insert(0, ROOT)
    import importlib
    rrt = importlib.import_module("rust-ray-tracer_amd")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    base = np.load(frame_path)
    n = rrt.tiles_per_rank(w, h, world) * 64
    mine = [torch.zeros(n, dtype=torch.int32) for _ in range(depth)]
    gathered = [torch.zeros(world * n, dtype=torch.int32) for _ in range(depth)] if rank == 0 else [None] * depth
    chunks = [[g[i * n:(i + 1) * n] for i in range(world)] if rank == 0 else None for g in gathered]
    work = [None] * depth
    pending = [None] * depth          # rank 0: (frame number) whose gather sits in the slot, not yet de-tiled
    ok = True

    def drain(b):
        nonlocal ok
        if work[b] is not None:
            work[b].wait(); work[b] = None
        if rank == 0 and pending[b] is not None:
            fb = rrt.detile_host(gathered[b].numpy().view(np.uint32), w, h, world)
            ok = ok and np.array_equal(fb, (base + np.uint32(pending[b])) & np.uint32(0xFFFFFF))
            pending[b] = None

    for i in range(steps):
        b = i % depth
        drain(b)                                                                   # slot free: previous gather done and consumed
        frame_i = (base + np.uint32(i)) & np.uint32(0xFFFFFF)
        mine[b].copy_(torch.from_numpy(_tile_major(frame_i, world, rank, rrt).view(np.int32).copy()).reshape(-1))   # "trace" this rank's tiles
        work[b] = dist.gather(mine[b], chunks[b], dst=0, async_op=True)
        if rank == 0:
            pending[b] = i
    for b in range(depth):
        drain(b)
    if rank == 0:
        np.save(os.path.join(out_dir, "ok.npy"), np.array([ok]))
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_gather_gloo_world2(rrt, teapot_oracle, tmp_path):
    w, h = 75, 37
    frame, _ = teapot_oracle.render(w, h)
    np.save(tmp_path / "frame.npy", frame)
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_pipelined_worker, args=(2, port, w, h, str(tmp_path / "frame.npy"), str(tmp_path), 7, 2), nprocs=2, join=True)
    assert bool(np.load(tmp_path / "ok.npy")[0])


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


def test_bench_supervisor_kills_a_hung_child_group():
    """bench.py's N > 1 supervisor (GPU-free parent): a child group that does not finish is killed as a process group at its timeout, and its last
    `[bench stage]` line and stderr tail are what gather_paths.<path>.error reports; a group that finishes hands back its one JSON line."""
    import importlib.util, sys, time
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    hang = "import sys, time; print('[bench stage] library gather: sleeping', file=sys.stderr, flush=True); time.sleep(600)"
    t0 = time.time()
    line, rc, tail, timed_out = bench.run_group([sys.executable, "-c", hang], dict(os.environ), 2.0)
    assert timed_out and line is None and rc != 0 and time.time() - t0 < 30 and any("[bench stage] library gather: sleeping" in t for t in tail)
    ok = "print('{\"metric\": \"m\", \"value\": 1}')"
    line, rc, tail, timed_out = bench.run_group([sys.executable, "-c", ok], dict(os.environ), 30.0)
    assert not timed_out and rc == 0 and line == '{"metric": "m", "value": 1}'
    assert bench._strip_opt(["--gpus", "8", "--gather", "lib", "--steps", "5", "--gather=torch"], "--gather") == ["--gpus", "8", "--steps", "5"]
