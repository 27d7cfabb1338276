#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native hot path (BASELINE.json: Mrays/s primary + frame ms, teapot @1920x1080).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--width 1920 --height 1080] [--scene assets/model2.obj] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one full frame of the scene (Scene::draw_scene, engine.rs:186-255) with the scene already resident in HBM and the
framebuffer left in HBM.  N = 1: one launch of the trace kernel writes the row-major framebuffer.  N > 1: the frame's 8x8-pixel
tiles are dealt round-robin to the ranks (tile k -> rank k % N), each rank traces its tiles, one RCCL gather over xGMI
collects the tile-major buffers on GPU 0, which de-tiles them to the row-major frame ("scaling": "strong" -- the frame is fixed).
Rank 0 prints ONE JSON line.  The `roofline` object prices the trace kernel against the f64 VECTOR peak (this path is VALU-bound:
no MFMA, almost no HBM traffic -- see DESIGN.md section 5) and carries the HBM figure the metric asks for as `hbm`.
`cpu_baseline` is the oracle (restatement of the reference's rayon CPU path; the Rust reference cannot be built here) timed on
the host cores -- a checker used as a reported baseline, never part of the product path.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_VECTOR_PEAK_TFLOPS = 78.6     # MI355X f64 vector (non-matrix) peak: 256 CU x 128 FLOP/clk x 2.4 GHz (FMA = 2 FLOP); = 1/2 of the fp32 vector 157.3 TF in MI355X_MICROARCH.md
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_flops(c: dict) -> float:
    """full-path Moller-Trumbore 52 flop, slab test 24, shading ~250 per shaded hit."""
    return 52.0 * c["tri_tests"] + 24.0 * c["aabb_tests"] + 250.0 * c["hits_shaded"]


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scene", default=os.path.join(ROOT, "assets", "model2.obj"))
    ap.add_argument("--pipeline-depth", type=int, default=4, help="N > 1: frames in flight (gather + de-tiling of a frame overlap the tracing of the next); 1 = none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU time of the bounded cpu_baseline sample")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON: RCCL prints a version banner and gloo its connection notes to fd 1 while the process group
    # comes up, so fd 1 points at stderr until the result is printed.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback in the product path)")
    # RRT_BENCH_REHEARSAL=1: developer rehearsal of the N > 1 code path on a ONE-GPU box -- every rank uses GPU 0 and the collective runs
    # over gloo (RCCL refuses two ranks on one device).  Never used by the driver; numbers from it are not benchmark results.
    rehearsal = os.environ.get("RRT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # RRT_BENCH_FORCE_DIST=1: developer check of the N > 1 choreography (RCCL gather, side stream, frames in flight) with ONE rank on one GPU
    multi = world > 1 or os.environ.get("RRT_BENCH_FORCE_DIST") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    rrt = importlib.import_module("rust-ray-tracer_amd")
    W, H = args.width, args.height
    if args.scene.startswith("soup"):   # soup100000 / soup1000000: the synthetic configs of BASELINE.json, generated on the spot (rank 0 writes the .obj)
        syn = importlib.import_module("rust-ray-tracer_amd.synthetic")
        n = int(args.scene[4:])
        if rank == 0:
            syn.ensure_soup(os.path.join(ROOT, "assets"), n, syn.SEED_100K if n == 100000 else syn.SEED_1M if n == 1000000 else 0x5EED0003)
        if world > 1:
            dist.barrier()
        args.scene = syn.ensure_soup(os.path.join(ROOT, "assets"), n, syn.SEED_100K if n == 100000 else syn.SEED_1M if n == 1000000 else 0x5EED0003)
    sd = rrt.parse_obj_file(args.scene)
    lights = rrt.default_lights()
    rt = rrt.RayTracer(sd, lights, rrt.DEFAULT_ORIGIN, device=local_rank)

    fb = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    DEPTH = max(1, args.pipeline_depth) if multi else 1
    if multi:
        # Frames in flight: frame i's gather (RCCL stream) and de-tiling (side stream on GPU 0) overlap the tracing of frame i+1, so each of the
        # DEPTH slots has its own tile buffer, gather buffer and completion handles.  Every frame still goes trace -> gather -> de-tile in full.
        tpr = rrt.tiles_per_rank(W, H, world)
        mine = [torch.zeros(tpr * 64, dtype=torch.int32, device="cuda") for _ in range(DEPTH)]
        gathered = [torch.zeros(world * tpr * 64, dtype=torch.int32, device="cuda") for _ in range(DEPTH)] if rank == 0 else [None] * DEPTH
        # final gather to GPU 0 (BASELINE.json north_star): grouped point-to-point sends, every peer on its own xGMI link -- not a ring
        chunks = [[g[i * tpr * 64:(i + 1) * tpr * 64] for i in range(world)] if rank == 0 else None for g in gathered]
        side = torch.cuda.Stream() if (rank == 0 and not rehearsal) else None       # de-tiling stream of GPU 0
        # each slot traces on its own stream: the next frame's waves fill the slots that the tail of the previous frame's launch leaves idle
        # (at N = 8 a rank's launch is only ~4 waves per wave slot deep)
        trace_streams = [torch.cuda.Stream() if not rehearsal else torch.cuda.current_stream() for _ in range(DEPTH)]
        gather_work = [None] * DEPTH                                                # outstanding gather of each slot
        detile_done = [torch.cuda.Event() for _ in range(DEPTH)]                    # slot's gather buffer has been de-tiled (GPU 0)
        detile_pending = [False] * DEPTH

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    # pre-bound launchers: one ctypes call per launch in the timed loop
    if not multi:
        launch_frame = rt.bind_render(fb, W, H)
    else:
        launch_tiles = [rt.bind_render_tiles(m, W, H, rank, world, stream=ts.cuda_stream) for m, ts in zip(mine, trace_streams)]
        launch_detile = [rt.bind_detile(g, fb, W, H, world, stream=(side.cuda_stream if side is not None else None)) for g in gathered] if rank == 0 else None
    frame_no = [0]
    set_stream = torch.cuda.set_stream
    default_stream = torch.cuda.current_stream()
    # plain fallback (no frames in flight: trace, gather, de-tile one after the other on the default stream), used with --pipeline-depth 0 or if the
    # pipelined step raises in its first warm-up call on any rank
    simple = [multi and args.pipeline_depth == 0]
    if multi:
        simple_tiles = rt.bind_render_tiles(mine[0], W, H, rank, world)
        simple_detile = rt.bind_detile(gathered[0], fb, W, H, world) if rank == 0 else None

    def simple_step(i: int | None) -> None:
        if i is not None: ev[i][0].record()
        simple_tiles()
        if i is not None: ev[i][1].record()
        dist.gather(mine[0], chunks[0], dst=0)
        if rank == 0:
            simple_detile()

    def step(i: int | None) -> None:
        if not multi:
            if i is not None: ev[i][0].record()
            launch_frame()
            if i is not None: ev[i][1].record()
            return
        if simple[0]:
            simple_step(i); return
        b = frame_no[0] % DEPTH; frame_no[0] += 1
        ts = trace_streams[b]
        set_stream(ts)                                       # (torch.cuda.set_stream, not the context manager: the host side of a step is on the critical path at N = 8)
        if gather_work[b] is not None:
            gather_work[b].wait()                            # the slot's previous gather has read mine[b] (orders the slot's stream after it; the host does not block on RCCL)
        if rank == 0 and detile_pending[b]:
            ts.wait_event(detile_done[b])                    # ... and its gather buffer has been consumed before the next gather (issued after this point) overwrites it
        if i is not None: ev[i][0].record(ts)
        launch_tiles[b]()
        if i is not None: ev[i][1].record(ts)
        gather_work[b] = dist.gather(mine[b], chunks[b], dst=0, async_op=True)   # RCCL's stream waits for the slot's stream
        if rank == 0:
            if side is None:                                 # rehearsal (gloo): synchronous
                gather_work[b].wait(); gather_work[b] = None
                launch_detile[b]()
            else:
                set_stream(side)
                gather_work[b].wait()                        # side stream waits for the gather; the tracing streams go on with the next frames
                launch_detile[b]()
                detile_done[b].record(side); detile_pending[b] = True

    def fence() -> None:
        if multi:
            set_stream(default_stream)
            for w in gather_work:
                if w is not None: w.wait()
            torch.cuda.synchronize()
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.synchronize()                                 # buffers were zero-filled on the default stream; the slots run on their own
    if multi and not simple[0]:
        ok = 1
        try:
            step(None); fence()
        except Exception as e:                               # noqa: BLE001 -- any failure of the pipelined choreography falls back to the plain step
            print(f"[bench rank {rank}] pipelined step failed ({type(e).__name__}: {e}); falling back to the plain step", file=sys.stderr, flush=True)
            ok = 0
        if world > 1:
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            set_stream(default_stream)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        if not ok:
            simple[0] = True
            for k in range(DEPTH): gather_work[k] = None; detile_pending[k] = False
            set_stream(default_stream); torch.cuda.synchronize()
    for _ in range(args.warmup):
        step(None)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    host_enqueue_ms = (time.perf_counter() - t0) / args.steps * 1e3   # host time to issue one step (launch + gather + de-tile calls), before any waiting
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))   # trace kernel only, HIP events on its launch stream
    if world > 1:
        kmax = torch.tensor([kernel_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
        kernel_ms = float(kmax.item())

    if rank == 0:
        traced_rows = 2 * (H // 2) - 1                       # y = -H/2 is computed by the reference but its pixels are rejected by put_pixel; not traced here
        rays_primary = 4 * (2 * (W // 2)) * traced_rows
        ms_per_step = elapsed / args.steps * 1e3
        frame = fb.cpu().numpy().view(np.uint32)

        # --- work counters for the algorithmic flop count (oracle counters; committed for the headline config, else scaled from the CPU sample)
        key = f"{os.path.basename(args.scene)}@{W}x{H}"
        counters, counters_src = None, None
        try:
            counters = json.load(open(os.path.join(ROOT, "bench_data", "work_counters.json")))["counters"].get(key)
            counters_src = "bench_data/work_counters.json (oracle, exact for this frame)" if counters else None
        except OSError:
            pass

        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import binding as ob                  # checker / reported baseline only
            pos, uv, nrm, mat = sd.triangles()
            osc = ob.OracleScene(pos, uv, nrm, mat, sd.materials(), sd.textures(), [(l.kind, l.intensity, (l.v.x, l.v.y, l.v.z)) for l in lights], (0.0, 2.0, -10.0))
            cores = os.cpu_count() or 1
            try:
                cores = len(os.sched_getaffinity(0))
            except AttributeError:
                pass
            tp = time.perf_counter(); osc.render(W // 8, H // 8, n_threads=cores); probe = time.perf_counter() - tp
            div = 1
            for cand in (1, 2, 3, 4, 6, 8):                   # largest sample (1/div of each dimension) expected to fit the time budget
                if probe * 64.0 / (cand * cand) <= args.cpu_seconds:
                    div = cand; break
                div = cand
            sw, sh = W // div, H // div
            tp = time.perf_counter(); ref, cnt = osc.render(sw, sh, n_threads=cores); cpu_s = time.perf_counter() - tp
            cpu_rays = cnt["rays_primary"]
            cpu = {"value": round(cpu_rays / cpu_s / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
                   "sample": f"oracle (f64 C restatement of the reference rayon path, -O3 -ffp-contract=off) rendering {os.path.basename(args.scene)} at {sw}x{sh} "
                             f"({cpu_rays} primary rays incl. the discarded row) in {cpu_s:.2f} s on {cores} threads; parallel over the rows of one 50-row chunk at a time "
                             f"with a barrier per chunk, as the reference's rayon loop (engine.rs:196-203), so at most 50 threads are busy at once",
                   "frame_ms_at_sample": round(cpu_s * 1e3, 1)}
            if div == 1:
                d = np.abs(np.stack([(frame >> s) & 255 for s in (16, 8, 0)], -1).astype(np.int64) - np.stack([(ref >> s) & 255 for s in (16, 8, 0)], -1).astype(np.int64))
                cpu["gpu_vs_cpu_max_channel_diff"] = int(d.max())
            if counters is None:
                scale = (rays_primary + 4 * 2 * (W // 2)) / cpu_rays
                counters = {k: v * scale for k, v in cnt.items()}
                counters_src = f"oracle counters of the {sw}x{sh} CPU sample scaled by ray count"

        roofline = None
        if counters is not None:
            # ALGORITHMIC work = what the reference's algorithm does for this frame (oracle counters): every triangle of every visited node's list
            # is tested, every child box slab-tested.  The kernel reaches the same pixels while skipping most of that work (result-preserving
            # cluster/subtree index), so algorithmic flop/s can exceed the machine peak; `executed` holds what the hardware really ran (rocprofv3
            # PMC of the same command, profiles/current_summary.json) and is the utilisation figure.
            flops = algorithmic_flops(counters)
            rays_all = counters["rays_primary"] + counters["rays_shadow"] + counters["rays_reflect"]
            per_rank = 1.0 / world
            ach = flops * per_rank / (kernel_ms * 1e-3) / 1e12
            hbm_bytes = rt.last_stats()["scene_bytes"] + 4.0 * W * H * per_rank
            prof = None
            try:
                prof = json.load(open(os.path.join(ROOT, "profiles", "current_summary.json")))
            except OSError:
                pass
            executed, traffic = None, None
            if prof and world == 1 and (W, H) == (1920, 1080) and os.path.basename(args.scene) == "model2.obj":
                pm = prof["pmc_avg_per_launch"]
                ex_flop = prof.get("executed_f64_flop_per_launch_upper_bound")
                prof_ms = float(prof["kernel_stats"]["AverageNs"]) * 1e-6
                executed = {"source": "profiles/current_summary.json (rocprofv3 --pmc, same command, one pass per counter group)",
                            "profiled_kernel_ms": round(prof_ms, 4), "valu_insts_per_launch": pm.get("SQ_INSTS_VALU"),
                            "f64_flop_per_launch_upper_bound": ex_flop,
                            "f64_tflops": round(ex_flop / (prof_ms * 1e-3) / 1e12, 3) if ex_flop else None,
                            "f64_frac_of_peak": round(ex_flop / (prof_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS, 4) if ex_flop else None,
                            # SQ_ACTIVE_INST_VALU counts quad-cycles summed over waves; 1024 SIMDs; clock taken as the 2.4 GHz peak (lower bound on busy)
                            "valu_busy_frac_at_2p4GHz": round(pm["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024 * 2.4e9 * prof_ms * 1e-3), 3) if "SQ_ACTIVE_INST_VALU" in pm else None,
                            "algorithmic_over_executed_flop": round(flops / ex_flop, 1) if ex_flop else None}
                hb = prof.get("hbm_bytes_per_launch")
                if hb:
                    traffic = hb["fetch_x2_gfx950_correction"] + hb["write"]   # MI355X_MICROARCH.md: FETCH_SIZE reads half the bytes of a wide stream on gfx950
            roofline = {"bound": "valu", "kernel": "render_kernel", "achieved": round(ach, 3), "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / FP64_VECTOR_PEAK_TFLOPS, 4), "traffic": traffic,
                        "note": "achieved = reference-algorithm flop / kernel time; > peak means the index skipped that work, see `executed` for hardware utilisation",
                        "kernel_ms": round(kernel_ms, 4), "algorithmic_flops_per_launch": flops * per_rank,
                        "flop_model": "52*tri_tests + 24*aabb_tests + 250*hits_shaded (SURVEY.md 8d), counts from the oracle",
                        "counters": {k: (int(v) if float(v).is_integer() else v) for k, v in counters.items() if k != "seconds_8_threads_container"},
                        "counters_source": counters_src, "rays_all_kinds": int(rays_all), "executed": executed,
                        "hbm": {"algorithmic_bytes_per_launch": hbm_bytes, "achieved": round(hbm_bytes / (kernel_ms * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBPS,
                                "unit": "GB/s", "frac": round(hbm_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6),
                                "note": "scene (uploaded once, L2/MALL resident) + framebuffer; the path is not HBM-bound"}}

        out = {"metric": "Mrays/s (primary) at 1920x1080, Utah teapot (model2.obj)" if (W, H) == (1920, 1080) else f"Mrays/s (primary) at {W}x{H}",
               "value": round(rays_primary / (elapsed / args.steps) / 1e6, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "reference scene assets/model2.obj (teapot+table+mirror, 6334 triangles), camera/lights of main.rs"
               if os.path.basename(args.scene) == "model2.obj" else "synthetic",
               "config": {"workload": f"{os.path.basename(args.scene)} {W}x{H}, 4 sub-samples/pixel, shadow rays + depth-5 mirror reflection, f64",
                          "rays_primary_per_frame": rays_primary, "partition": "single launch" if world == 1 else f"8x8-pixel tiles round-robin over {world} GPUs + " + "RCCL gather to GPU 0, " + ("one frame at a time" if simple[0] else f"{DEPTH} frames in flight (one stream per slot)"),
                          "octree_nodes": sd.info["n_nodes"], "triangles": sd.info["n_tris"]},
               "frame_ms": round(ms_per_step, 4), "kernel_ms": round(kernel_ms, 4), "host_enqueue_ms_per_step": round(host_enqueue_ms, 4), **({"rehearsal": "gloo on one GPU: NOT a benchmark result"} if rehearsal else {}),
               "frame_checksum": int(np.bitwise_xor.reduce(frame.ravel().astype(np.uint64) * np.arange(1, frame.size + 1, dtype=np.uint64)))}
        if roofline is not None:
            out["roofline"] = roofline
        if cpu is not None:
            out["cpu_baseline"] = cpu
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)

    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
