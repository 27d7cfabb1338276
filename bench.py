#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native hot path (BASELINE.json: Mrays/s primary + frame ms, teapot @1920x1080).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--width 1920 --height 1080] [--scene assets/model2.obj | soup100000 | soup1000000]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    (a plain `python bench.py --gpus N` with no WORLD_SIZE in the environment starts those N ranks itself and relays rank 0's JSON line)

A step = one full frame of the scene (Scene::draw_scene, engine.rs:186-255) with the scene already resident in HBM and the framebuffer left in
HBM.  N = 1: one launch of the trace kernel writes the row-major framebuffer.  N > 1: the frame's 8x8-pixel tiles are dealt round-robin to the
ranks (tile k -> rank k % N), each rank traces its tiles, one RCCL gather over xGMI collects the tile-major buffers on GPU 0, which de-tiles them
into the row-major frame ("scaling": "strong" -- the frame is fixed).  The documented configs[3] line is `--gpus 8 --width 3840 --height 2160`.
Rank 0 prints ONE JSON line.

`roofline` prices the trace kernel against the f64 VECTOR peak (the path is VALU-bound: no MFMA, almost no HBM traffic -- DESIGN.md section 4):
`achieved` = f64 flop the hardware EXECUTED per launch (rocprofv3 SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 of the same command, committed under
profiles/ and stamped with the sha256 of the kernel sources) / the kernel's launch duration measured live with HIP events; `frac` <= 1.  The
reference algorithm's own flop count (oracle counters; the index skips most of that work) is reported beside it as `work_skipped_vs_reference`.
`cpu_baseline` is the oracle (restatement of the reference's rayon CPU path; the Rust reference cannot be built here) timed on the host cores --
a checker used as a reported baseline, never part of the product path.
"""
from __future__ import annotations

import argparse
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_VECTOR_PEAK_TFLOPS = 78.6     # MI355X f64 vector (non-matrix) peak: 256 CU x 128 FLOP/clk x 2.4 GHz (FMA = 2 FLOP); = 1/2 of the fp32 vector 157.3 TF in MI355X_MICROARCH.md
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
KERNEL_SOURCES = ("rust-ray-tracer_amd/csrc/render.hip", "rust-ray-tracer_amd/csrc/clusters.cpp", "rust-ray-tracer_amd/csrc/device_scene.hpp",
                  "rust-ray-tracer_amd/csrc/Makefile")   # the Makefile carries the code-generation switches the kernels are built with


def kernel_source_sha256() -> str:
    """Identity of the code the PMC figures under profiles/ were measured on (tools/summarize_profiles.py stamps the same hash)."""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def algorithmic_flops(c: dict) -> float:
    """full-path Moller-Trumbore 52 flop, slab test 24, shading ~250 per shaded hit."""
    return 52.0 * c["tri_tests"] + 24.0 * c["aabb_tests"] + 250.0 * c["hits_shaded"]


def _strip_opt(argv, name):
    out, skip = [], False
    for a in argv:
        if skip: skip = False; continue
        if a == name: skip = True; continue
        if a.startswith(name + "="): continue
        out.append(a)
    return out


def run_group(cmd, env, timeout):
    """One group of fresh child processes under its own timeout.  Returns (json_line | None, rc | None, stderr tail, timed_out).  The children run in a
    process group of their own; a group that does not finish is killed as a group (this process never touches the GPU, so it can always do that)."""
    import signal, threading
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    line, tail = [None], []

    def pump_out():
        for out in p.stdout:
            t = out.strip()
            if t.startswith("{") and '"metric"' in t: line[0] = t
            elif t: tail.append(t); del tail[:-40]

    def pump_err():
        for out in p.stderr:
            t = out.rstrip()
            if t:
                tail.append(t); del tail[:-40]
                print(t, file=sys.stderr, flush=True)
    th = [threading.Thread(target=pump_out, daemon=True), threading.Thread(target=pump_err, daemon=True)]
    for t in th: t.start()
    timed_out = False
    try:
        rc = p.wait(timeout=timeout)
    except subprocess.TimeoutExpired:
        timed_out = True
        try:
            os.killpg(p.pid, signal.SIGKILL)                  # exactly the process group started above
        except ProcessLookupError:
            pass
        rc = p.wait()
    for t in th: t.join(timeout=5)
    return line[0], rc, tail, timed_out


def supervise(args, argv) -> int:
    """N > 1 (or RRT_BENCH_FORCE_DIST=1): this process stays GPU-free and runs every gather path as its OWN group of fresh child processes, one group
    after the other, each under its own timeout -- `--gather auto`: the torch.distributed.gather choreography, then the library's RCCL gather.  A group
    that does not finish is killed by this process; its last stage and stderr tail go into gather_paths.<path>.error and the other path's line is
    printed.  rc != 0 only if no path finished.  Works in both launch modes: started plainly (`python bench.py --gpus N`: each group is one
    torch.distributed.run) or as a rank under a launcher (the driver's `python -m torch.distributed.run ... bench.py --gpus N`: every rank supervises
    its own children, which rendezvous on MASTER_PORT + 11 + k)."""
    under_launcher = "WORLD_SIZE" in os.environ
    rank = int(os.environ.get("RANK", "0"))
    force_one = os.environ.get("RRT_BENCH_FORCE_DIST") == "1" and args.gpus == 1
    paths = ["torch", "lib"] if args.gather == "auto" else [args.gather]
    base = _strip_opt(argv, "--gather")
    lines, info = {}, {}
    for k, path in enumerate(paths):
        timeout = args.lib_timeout if path == "lib" else args.torch_timeout
        env = dict(os.environ)
        child = [os.path.abspath(__file__)] + base + ["--gather", path, "--child"]
        if under_launcher:
            env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 11 + k)
            env.pop("TORCHELASTIC_USE_AGENT_STORE", None)      # the children rendezvous among themselves: rank 0's child hosts the store on the new port
            cmd = [sys.executable] + child
        elif force_one:
            env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29533")) + k)
            cmd = [sys.executable] + child
        else:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port)] + child
        t0 = time.time()
        line, rc, tail, timed_out = run_group(cmd, env, timeout)
        stage = next((t.split("[bench stage]", 1)[1].strip() for t in reversed(tail) if "[bench stage]" in t), None)
        ok = (rc == 0) and (line is not None or rank != 0)
        if ok:
            if line is not None: lines[path] = json.loads(line)
            info[path] = {"ok": True}
        else:
            why = f"timed out after {timeout:.0f} s and was killed by the supervisor" if timed_out else f"exited with rc {rc}" + ("" if line else " and printed no result")
            info[path] = {"ok": False, "error": f"{path} gather group {why}; last stage: {stage or 'unknown'}; stderr tail: {' | '.join(tail[-4:])[-600:]}", "seconds": round(time.time() - t0, 1)}
            print(f"[bench supervisor rank {rank}] {info[path]['error']}", file=sys.stderr, flush=True)
    finished = [pth for pth in paths if info[pth]["ok"]]
    if rank == 0 and finished:
        summary = {}
        for pth in paths:
            o = lines.get(pth)
            summary[pth] = ({"value": o["value"], "ms_per_step": o["ms_per_step"], "gather_ms": o.get("gather_ms"), "host_enqueue_ms_per_step": o["host_enqueue_ms_per_step"],
                             "frames_in_flight": o.get("frames_in_flight"), "frame_checksum": o["frame_checksum"]} if o else {"error": info[pth].get("error")})
        cands = [lines[pth] for pth in finished if pth in lines]
        best = max(cands, key=lambda o: o["value"])
        if "torch" in lines and "lib" in lines and lines["lib"]["frame_checksum"] != lines["torch"]["frame_checksum"]:
            best = lines["torch"]; summary["lib"]["error"] = "frame differs from the torch path's"
        best["gather_paths"] = summary
        print(json.dumps(best), flush=True)
    return 0 if finished else 1


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scene", default=os.path.join(ROOT, "assets", "model2.obj"))
    ap.add_argument("--pipeline-depth", type=int, default=4, help="N > 1: frames in flight (gather + de-tiling of a frame overlap the tracing of the next); 0 = plain one-frame-at-a-time step")
    ap.add_argument("--gather", choices=("auto", "lib", "torch"), default="auto", help="N > 1: `lib` = the library's own RCCL gather (rrt_dist_create / rrt_multi_enqueue: partition, "
                    "gather and de-tiling behind the C ABI); `torch` = the same choreography issued from here with torch.distributed.gather; `auto` = K steps of each, "
                    "each path in its own group of fresh child processes under its own timeout (supervise()), the faster one reported (both in `gather_paths`)")
    ap.add_argument("--lib-timeout", type=float, default=180.0, help="N > 1: seconds the library-gather child group may take before the supervisor kills it")
    ap.add_argument("--torch-timeout", type=float, default=300.0, help="N > 1: seconds the torch-gather child group may take before the supervisor kills it")
    ap.add_argument("--child", action="store_true", help="internal: this process is one rank of ONE gather path's child group (started by the supervisor)")
    ap.add_argument("--walk", choices=("auto", "lane", "bundle", "ray"), default="auto", help="traversal variant: auto = the library measures all three on the second frame of a size (the product default); the others force one (developer A/B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-first-frame", action="store_true", help="skip the cold time-to-first-frame measurement (first_frame_ms)")
    ap.add_argument("--no-host-fb", action="store_true", help="skip the boundary-inclusive rrt_render timings (frame_ms_host_fb)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU time of the bounded cpu_baseline leg (3 samples)")
    args = ap.parse_args()

    if (args.gpus > 1 or os.environ.get("RRT_BENCH_FORCE_DIST") == "1") and not args.child:
        raise SystemExit(supervise(args, sys.argv[1:]))            # GPU-free from here on: every gather path runs in fresh child processes
    if args.child and args.gather == "auto":
        raise SystemExit("bench.py --child needs an explicit --gather torch|lib")

    def stage(msg):
        print(f"[bench stage] {msg}", file=sys.stderr, flush=True)

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
    scene_name = os.path.basename(args.scene)
    if args.scene.startswith("soup"):   # soup100000 / soup1000000: the synthetic configs of BASELINE.json, generated on the spot (rank 0 writes the .obj)
        syn = importlib.import_module("rust-ray-tracer_amd.synthetic")
        n = int(args.scene[4:])
        seed = syn.SEED_100K if n == 100000 else syn.SEED_1M if n == 1000000 else 0x5EED0003
        if rank == 0:
            syn.ensure_soup(os.path.join(ROOT, "assets"), n, seed)
        if world > 1:
            dist.barrier()
        args.scene = syn.ensure_soup(os.path.join(ROOT, "assets"), n, seed)
    t_setup = time.perf_counter()
    sd = rrt.parse_obj_file(args.scene)
    lights = rrt.default_lights()
    rt = rrt.RayTracer(sd, lights, rrt.DEFAULT_ORIGIN, device=local_rank, box_filter=None if args.walk == "auto" else args.walk)
    setup_wall_ms = (time.perf_counter() - t_setup) * 1e3
    setup = {k: round(v, 2) for k, v in rt.setup_times().items()}
    setup["total_ms"] = round(sum(v for k, v in setup.items() if k not in ("create_ms", "gpu_setup")), 2); setup["wall_ms_incl_binding"] = round(setup_wall_ms, 2)

    fb = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    set_stream = torch.cuda.set_stream
    default_stream = torch.cuda.current_stream()
    pipeline_fallback, pipeline_error = False, None
    gather_ev = []

    if not multi:
        DEPTH = 1
        launch_frame = rt.bind_render(fb, W, H)

        def step(i):
            if i is not None: ev[i][0].record()
            launch_frame()
            if i is not None: ev[i][1].record()

        def fence():
            torch.cuda.synchronize()
    else:
        # ---- N > 1.  Everything that can fail locally (buffers, streams, bound launchers, the filter tuning inside the first launch) is set up
        # and exercised BEFORE any collective of the pipelined step is issued; the ranks then agree (one all_reduce) on pipelined vs plain.  A
        # failure after that point is not recoverable without risking mismatched collectives: the rank tears the process group down and exits non-zero.
        tpr = rrt.tiles_per_rank(W, H, world)
        want_pipeline = args.pipeline_depth > 0
        DEPTH = max(1, args.pipeline_depth)
        lib_mg = None
        lib_state = {"stage": "not started", "error": None}

        def lib_init():
            """The library's own gather: (a) local -- bind RCCL inside the library (dlopen), no communication; the ranks agree; (b) collective --
            ncclCommInitRank with rank 0's id."""
            ok, err, uid = 1, None, None
            try:
                uid = rrt.MultiGpu.unique_id()
            except Exception as e:                                                       # noqa: BLE001
                ok = 0; err = f"rank {rank}: {type(e).__name__}: {e}"
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                lib_state["error"] = err or "another rank could not bind RCCL inside the library"
                return None
            idt = torch.frombuffer(bytearray(uid), dtype=torch.uint8).cuda()
            dist.broadcast(idt, src=0)
            stage("library gather: ncclCommInitRank (rrt_dist_create)")
            return rrt.MultiGpu.dist(rt, rank, world, bytes(idt.cpu().numpy().tobytes()), frames_in_flight=DEPTH)   # collective
        ok = 1
        try:
            # Frames in flight: frame i's gather (RCCL stream) and de-tiling (side stream on GPU 0) overlap the tracing of frame i+1, so each of
            # the DEPTH slots has its own tile buffer, gather buffer and completion handles.  Every frame still goes trace -> gather -> de-tile in full.
            mine = [torch.zeros(tpr * 64, dtype=torch.int32, device="cuda") for _ in range(DEPTH)]
            gathered = [torch.zeros(world * tpr * 64, dtype=torch.int32, device="cuda") for _ in range(DEPTH)] if rank == 0 else [None] * DEPTH
            # final gather to GPU 0 (BASELINE.json north_star): grouped point-to-point sends, every peer on its own xGMI link -- not a ring
            chunks = [[g[i * tpr * 64:(i + 1) * tpr * 64] for i in range(world)] if rank == 0 else None for g in gathered]
            side = torch.cuda.Stream() if (rank == 0 and not rehearsal) else None       # de-tiling stream of GPU 0
            # each slot traces on its own stream: the next frame's waves fill the slots that the tail of the previous frame's launch leaves idle
            trace_streams = [torch.cuda.Stream() if not rehearsal else torch.cuda.current_stream() for _ in range(DEPTH)]
            launch_tiles = [rt.bind_render_tiles(m, W, H, rank, world, stream=ts.cuda_stream) for m, ts in zip(mine, trace_streams)]
            launch_detile = [rt.bind_detile(g, fb, W, H, world, stream=(side.cuda_stream if side is not None else None)) for g in gathered] if rank == 0 else None
            simple_tiles = rt.bind_render_tiles(mine[0], W, H, rank, world)
            simple_detile = rt.bind_detile(gathered[0], fb, W, H, world) if rank == 0 else None
            for lt in launch_tiles: lt()                                                 # local launches only: tunes the filter variant, proves the kernels run
            simple_tiles()
            if rank == 0: launch_detile[0](); simple_detile()
            torch.cuda.synchronize()
        except Exception as e:                                                           # noqa: BLE001
            ok = 0; pipeline_error = f"rank {rank}: {type(e).__name__}: {e}"
            print(f"[bench rank {rank}] local set-up of the N > 1 step failed: {pipeline_error}", file=sys.stderr, flush=True)
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            dist.destroy_process_group()
            raise SystemExit(f"bench.py: the N > 1 step could not be set up on every rank ({pipeline_error or 'another rank failed'})")
        gather_work = [None] * DEPTH                                                # outstanding gather of each slot
        detile_done = [torch.cuda.Event() for _ in range(DEPTH)]                    # slot's gather buffer has been de-tiled (GPU 0)
        detile_pending = [False] * DEPTH
        frame_no = [0]
        simple = [not want_pipeline]
        gather_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)] if rank == 0 else []

        def simple_step(i):
            if i is not None: ev[i][0].record()
            simple_tiles()
            if i is not None: ev[i][1].record()
            dist.gather(mine[0], chunks[0], dst=0)
            if rank == 0:
                simple_detile()
                if i is not None: gather_ev[i][0].record(); gather_ev[i][1].record()   # (plain step: gather and de-tile are in line; no separate figure)

        def pipelined_step(i):
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
                    if i is not None: gather_ev[i][0].record(side)
                    gather_work[b].wait()                        # side stream waits for the gather; the tracing streams go on with the next frames
                    launch_detile[b]()
                    if i is not None: gather_ev[i][1].record(side)
                    detile_done[b].record(side); detile_pending[b] = True

        def torch_step(i):
            try:
                simple_step(i) if simple[0] else pipelined_step(i)
            except Exception as e:                               # noqa: BLE001 -- collectives are in flight: no safe fallback from here
                print(f"[bench rank {rank}] step failed after collectives were issued ({type(e).__name__}: {e}); aborting", file=sys.stderr, flush=True)
                stage(f"torch gather path: step failed after collectives were issued ({type(e).__name__})")
                raise SystemExit(3)                              # non-zero; should teardown hang, the supervisor's timeout ends the group

        def torch_fence():
            set_stream(default_stream)
            for w in gather_work:
                if w is not None: w.wait()
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

        step, fence = torch_step, torch_fence
        # one full pipelined step per slot, then agree again: a rank whose pipelined choreography misbehaves WITHOUT raising (wrong frame) cannot be
        # detected here, but a rank that sees an error state on its streams can still vote for the plain step before the timed region
        if not simple[0] and args.gather != "lib":
            for _ in range(DEPTH): step(None)
            fence()
            ok = 1
            try:
                torch.cuda.synchronize()
            except Exception as e:                               # noqa: BLE001
                ok = 0; pipeline_error = f"rank {rank}: {type(e).__name__}: {e}"
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                pipeline_fallback = True; simple[0] = True
                for k in range(DEPTH): gather_work[k] = None; detile_pending[k] = False

    def measure(step, fence, lib=None):
        """W untimed warm-up steps, then exactly K steps between two fences (barrier + synchronize); MAX over ranks."""
        torch.cuda.synchronize()
        for _ in range(args.warmup):
            step(None)
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        host_enqueue_ms = (time.perf_counter() - t0) / args.steps * 1e3   # host time to issue one step (launch + gather + de-tile calls), before any waiting
        fence()
        elapsed = time.perf_counter() - t0
        if lib is not None:
            kernel_ms = float(rt.last_stats()["kernel_ms"])         # library path: HIP events around the rank's last trace launch, on its slot's stream
        else:
            kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))   # trace kernel only, HIP events on its launch stream
        m = {"elapsed": elapsed, "kernel_ms": kernel_ms, "kernel_ms_min": kernel_ms, "kernel_ms_max": kernel_ms, "gather_ms": None, "host_enqueue_ms": host_enqueue_ms}
        if multi:
            t = torch.tensor([elapsed, kernel_ms, -kernel_ms], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            m["elapsed"], m["kernel_ms_max"], m["kernel_ms_min"] = float(t[0]), float(t[1]), -float(t[2])
            m["kernel_ms"] = m["kernel_ms_max"]
            if rank == 0 and lib is not None:
                m["gather_ms"] = lib.last_gather_ms()               # rank 0's stream: own tiles traced -> frame de-tiled (wait for the slowest peer + gather + de-tile), last frame
            elif rank == 0 and gather_ev and not simple[0] and not rehearsal:
                m["gather_ms"] = float(np.mean([a.elapsed_time(b) for a, b in gather_ev]))   # side stream: wait for the RCCL gather of the slot + de-tile
        if rank == 0:
            m["frame"] = fb.cpu().numpy().view(np.uint32).copy()
        return m

    def build_out(m, gather_label):
        traced_rows = 2 * (H // 2) - 1                       # y = -H/2 is computed by the reference but its pixels are rejected by put_pixel; not traced here
        rays_primary = 4 * (2 * (W // 2)) * traced_rows
        elapsed, kernel_ms, kernel_ms_min, kernel_ms_max, gather_ms, host_enqueue_ms = m["elapsed"], m["kernel_ms"], m["kernel_ms_min"], m["kernel_ms_max"], m["gather_ms"], m["host_enqueue_ms"]
        in_lib = gather_label == "lib"
        ms_per_step = elapsed / args.steps * 1e3
        frame = m["frame"]
        workload_key = f"{'soup' + str(rt.info['n_tris']) if scene_name.startswith('soup') else scene_name}@{W}x{H}"

        # --- boundary-inclusive frame: rrt_render into a HOST framebuffer, as the Rust host's Canvas.buffer receives it (engine.rs:127,246-250)
        host_fb = None
        if world == 1 and not multi and not args.no_host_fb:
            n_host = max(5, min(50, args.steps))
            buf = np.empty((H, W), np.uint32)
            L = rrt.lib(); import ctypes as C
            ptr = buf.ctypes.data_as(C.POINTER(C.c_uint32))
            def timed():
                for _ in range(3): L.rrt_render(rt._h, W, H, ptr)
                tt = time.perf_counter()
                for _ in range(n_host): L.rrt_render(rt._h, W, H, ptr)
                return (time.perf_counter() - tt) / n_host * 1e3
            pageable = timed()
            same = bool(np.array_equal(buf, frame))
            L.rrt_host_buffer_register(C.c_void_p(buf.ctypes.data), buf.nbytes)
            try:
                registered = timed()
                same = same and bool(np.array_equal(buf, frame))
            finally:
                L.rrt_host_buffer_unregister(C.c_void_p(buf.ctypes.data))
            host_fb = {"frame_ms_host_fb": round(registered, 4), "frame_ms_host_fb_pageable": round(pageable, 4), "frames": n_host, "identical_to_device_frame": same,
                       "note": "wall time of the blocking rrt_render(rt, w, h, host_fb): kernel + device->host copy; `frame_ms_host_fb` with the caller's buffer page-locked "
                               "(rrt_host_buffer_register: one DMA), `..._pageable` through the library's pinned staging + pipelined host copy"}

        # --- time to FIRST frame (the reference renders one frame per run, main.rs:59-76): a host that has parsed the scene itself hands the triangle
        # arrays over (rrt_model_from_arrays), creates the raytracer (upload + octree + index built on the GPU) and renders one frame into a pageable
        # host framebuffer -- cold scene, cold variant choice, warm process (the HIP context exists; hip_init_ms is a one-off of the process).
        first = None
        if world == 1 and not multi and not args.no_first_frame:
            pos, uv, nrm, mat = sd.triangles(); mats_, texs_ = sd.materials(), sd.textures()
            def cold(host_setup):
                t0 = time.perf_counter()
                sd2 = rrt.SceneData.from_arrays(pos, uv, nrm, mat, mats_, texs_)
                t1 = time.perf_counter()
                rt2 = rrt.RayTracer(sd2, lights, rrt.DEFAULT_ORIGIN, device=local_rank, host_setup=host_setup)
                t2 = time.perf_counter()
                f = rt2.render(W, H)
                t3 = time.perf_counter()
                st = rt2.setup_times()
                return {"first_frame_ms": round((t3 - t0) * 1e3, 2), "model_from_arrays_ms": round((t1 - t0) * 1e3, 2), "raytracer_create_ms": round((t2 - t1) * 1e3, 2),
                        "first_render_ms": round((t3 - t2) * 1e3, 2), "octree_ms": round(st["octree_ms"], 2), "index_ms": round(st["index_ms"], 2), "upload_ms": round(st["upload_ms"], 2),
                        "variant": rrt.VARIANT_NAMES[rt2.last_stats()["filter_variant"]], "identical_to_steady_frame": bool(np.array_equal(f, frame))}
            def cold_direct():
                t0 = time.perf_counter()
                rt2 = rrt.RayTracer.from_arrays(pos, uv, nrm, mat, mats_, texs_, lights, rrt.DEFAULT_ORIGIN, device=local_rank)
                t1 = time.perf_counter()
                f = rt2.render(W, H)
                t2 = time.perf_counter()
                return {"first_frame_ms": round((t2 - t0) * 1e3, 2), "raytracer_create_from_arrays_ms": round((t1 - t0) * 1e3, 2), "first_render_ms": round((t2 - t1) * 1e3, 2),
                        "identical_to_steady_frame": bool(np.array_equal(f, frame))}
            def cold_from_file():
                t0 = time.perf_counter()
                sd2 = rrt.parse_obj_file(args.scene)
                t1 = time.perf_counter()
                rt2 = rrt.RayTracer(sd2, lights, rrt.DEFAULT_ORIGIN, device=local_rank)
                t2 = time.perf_counter()
                f = rt2.render(W, H)
                t3 = time.perf_counter()
                st = rt2.setup_times()
                return {"first_frame_ms": round((t3 - t0) * 1e3, 2), "parse_obj_file_ms": round((t1 - t0) * 1e3, 2), "of_it_texture_decode_ms": round(st["texture_ms"], 2),
                        "raytracer_create_ms": round((t2 - t1) * 1e3, 2), "first_render_ms": round((t3 - t2) * 1e3, 2), "identical_to_steady_frame": bool(np.array_equal(f, frame))}
            runs = [cold(False) for _ in range(3)]
            first = dict(sorted(runs, key=lambda r: r["first_frame_ms"])[1], runs_ms=[r["first_frame_ms"] for r in runs])
            first["host_setup"] = cold(True)
            first["without_model"] = sorted([cold_direct() for _ in range(3)], key=lambda r: r["first_frame_ms"])[1]
            first["from_file"] = sorted([cold_from_file() for _ in range(3)], key=lambda r: r["first_frame_ms"])[1]
            first["note"] = ("median of 3 cold runs: rrt_model_from_arrays -> rrt_raytracer_create (GPU set-up: pinned-staging upload, octree.rs:41-241 level-parallel on the device, index, "
                             "records) -> first rrt_render into a pageable host framebuffer; `host_setup` = the same with RRT_FLAG_HOST_SETUP (round-2 path), once; "
                             "`without_model` = rrt_raytracer_create_from_arrays -> first rrt_render (the host's arrays uploaded from where they lie, no rrt_model copy), median of 3; "
                             "`from_file` = rrt_model_load_obj (.obj/.mtl parse + JPEG decode on the host) -> rrt_raytracer_create -> first rrt_render: the reference's whole run, median of 3")

        # --- work counters for the reference algorithm's flop count (oracle counters; committed for the headline config, else scaled from the CPU sample)
        counters, counters_src = None, None
        try:
            counters = json.load(open(os.path.join(ROOT, "bench_data", "work_counters.json")))["counters"].get(workload_key)
            counters_src = "bench_data/work_counters.json (oracle, exact for this frame)" if counters else None
        except OSError:
            pass

        cpu = None
        if world == 1 and not multi and not args.no_cpu_baseline:
            from oracle import binding as ob                  # checker / reported baseline only
            pos, uv, nrm, mat = sd.triangles()
            osc = ob.OracleScene(pos, uv, nrm, mat, sd.materials(), sd.textures(), [(l.kind, l.intensity, (l.v.x, l.v.y, l.v.z)) for l in lights], (0.0, 2.0, -10.0))
            cores = os.cpu_count() or 1
            try:
                cores = len(os.sched_getaffinity(0))
            except AttributeError:
                pass
            tp = time.perf_counter(); osc.render(W // 8, H // 8, n_threads=cores); probe = time.perf_counter() - tp
            per_sample = args.cpu_seconds / 3.0
            div = 8
            for cand in (1, 2, 3, 4, 6, 8):                   # largest sample (1/div of each dimension) expected to fit the time budget
                if probe * 64.0 / (cand * cand) <= per_sample:
                    div = cand; break
            sw, sh = W // div, H // div
            secs = []
            for _ in range(3):
                tp = time.perf_counter(); ref, cnt = osc.render(sw, sh, n_threads=cores); secs.append(time.perf_counter() - tp)
            cpu_s = float(np.median(secs))
            cpu_rays = cnt["rays_primary"]
            cpu = {"value": round(cpu_rays / cpu_s / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
                   "sample": f"oracle (f64 C restatement of the reference rayon path, -O3 -ffp-contract=off) rendering {scene_name} at {sw}x{sh} "
                             f"({cpu_rays} primary rays incl. the discarded row): median of 3 samples ({', '.join(f'{s:.2f}' for s in secs)} s; the /8-size probe before them is the warm-up) "
                             f"on {cores} threads; parallel over the rows of one 50-row chunk at a time with a barrier per chunk, as the reference's rayon loop "
                             f"(engine.rs:196-203), so at most 50 threads are busy at once",
                   "frame_ms_at_sample": round(cpu_s * 1e3, 1), "samples": 3}
            if div == 1:
                d = np.abs(np.stack([(frame >> s) & 255 for s in (16, 8, 0)], -1).astype(np.int64) - np.stack([(ref >> s) & 255 for s in (16, 8, 0)], -1).astype(np.int64))
                cpu["gpu_vs_cpu_max_channel_diff"] = int(d.max())
                cpu["gpu_vs_cpu_pixels_with_nonzero_diff"] = int((d.max(-1) > 0).sum())      # of the whole frame (stated tolerance: 1 per channel, for pow)
            if counters is None:
                scale = (rays_primary + 4 * 2 * (W // 2)) / cpu_rays
                counters = {k: v * scale for k, v in cnt.items()}
                counters_src = f"oracle counters of the {sw}x{sh} CPU sample scaled by ray count"

        # --- roofline: executed f64 flop (rocprofv3 PMC of this workload, committed under profiles/, valid only for the kernel sources it was measured on)
        per_rank = 1.0 / world
        prof, prof_reason = None, None
        try:
            allp = json.load(open(os.path.join(ROOT, "profiles", "current_summary.json")))
            if allp.get("source_sha256") != kernel_source_sha256():
                prof_reason = "profiles/current_summary.json was measured on different kernel sources (source_sha256 mismatch): re-run tools/collect_profiles.sh"
            elif world != 1:
                prof_reason = "PMC profile is of the single-GPU launch"
            else:
                prof = allp.get("workloads", {}).get(workload_key)
                if prof is None:
                    prof_reason = f"no PMC profile of {workload_key} under profiles/"
        except (OSError, ValueError) as e:
            prof_reason = f"profiles/current_summary.json unreadable: {e}"
        executed, traffic, achieved, frac, valu_issue = None, None, None, None, None
        if prof:
            pm = prof["pmc_avg_per_launch"]
            ex_flop = prof.get("executed_f64_flop_per_launch_upper_bound")
            prof_ms = float(prof["kernel_stats"]["AverageNs"]) * 1e-6
            achieved = ex_flop / (kernel_ms * 1e-3) / 1e12 if ex_flop else None
            frac = achieved / FP64_VECTOR_PEAK_TFLOPS if achieved else None
            # SQ_ACTIVE_INST_VALU counts quad-cycles summed over waves; 1024 SIMDs; clock taken as the 2.4 GHz peak (lower bound on busy)
            valu_issue = pm["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024 * 2.4e9 * prof_ms * 1e-3) if "SQ_ACTIVE_INST_VALU" in pm else None
            executed = {"source": f"profiles/current_summary.json [{workload_key}] (rocprofv3 --pmc, same command, one pass per counter group; kernel sources sha256 {allp['source_sha256'][:12]})",
                        "profiled_kernel_ms": round(prof_ms, 4), "rocprof_vs_live_kernel_ms": round(prof_ms / kernel_ms, 3), "valu_insts_per_launch": pm.get("SQ_INSTS_VALU"),
                        "f64_flop_per_launch": ex_flop,
                        "f64_valu_inst_share": round((pm.get("SQ_INSTS_VALU_ADD_F64", 0) + pm.get("SQ_INSTS_VALU_MUL_F64", 0) + pm.get("SQ_INSTS_VALU_FMA_F64", 0) + pm.get("SQ_INSTS_VALU_TRANS_F64", 0)) / pm["SQ_INSTS_VALU"], 3) if pm.get("SQ_INSTS_VALU") else None}
            hb = prof.get("hbm_bytes_per_launch")
            if hb:
                traffic = hb["fetch_x2_gfx950_correction"] + hb["write"]   # MI355X_MICROARCH.md: FETCH_SIZE reads half the bytes of a wide stream on gfx950
        hbm_bytes = rt.last_stats()["scene_bytes"] + 4.0 * W * H * per_rank
        roofline = {"bound": "valu", "kernel": "render_kernel", "achieved": round(achieved, 3) if achieved else None, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(frac, 4) if frac else None, "traffic": traffic, "valu_issue_frac": round(valu_issue, 3) if valu_issue else None,
                    "kernel_ms": round(kernel_ms, 4),
                    "note": "achieved = f64 flop EXECUTED per launch (rocprofv3 SQ_INSTS_VALU_{ADD,MUL,FMA x2,TRANS}_F64 x 64 lanes) / live kernel time; peak = f64 vector peak with FMA, "
                            "which the reference's unfused arithmetic (-ffp-contract=off, required for parity) can reach at most half of.  The fraction counts f64 work EXECUTED: "
                            "removing f64 work that decides nothing lowers it while the frame gets faster (round 3: teapot 9.49 -> 7.01 GFLOP per frame, 0.873 -> 0.841 ms, "
                            "0.139 -> 0.107; DESIGN.md section 4) -- valu_issue_frac beside it is the pipe's actual occupancy",
                    "executed": executed, "executed_unavailable_reason": prof_reason,
                    "hbm": {"algorithmic_bytes_per_launch": hbm_bytes, "achieved": round(hbm_bytes / (kernel_ms * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBPS,
                            "unit": "GB/s", "frac": round(hbm_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6),
                            "measured_bytes_per_launch": traffic, "measured_over_algorithmic": round(traffic / hbm_bytes, 2) if traffic else None,
                            "note": "algorithmic = scene (uploaded once, L2/MALL resident) + framebuffer; the path is not HBM-bound.  The excess over it is write traffic: a wave covers 4x4 pixels and stores four "
                                    "16-byte row segments (quarter lines), plus the reflection stack in scratch; an 8x2 footprint (32-byte segments) was measured: +1 % frame time on the teapot, "
                                    "nothing on the soups (profiles/r03_ab_wave_footprint.txt) -- not adopted"}}
        if counters is not None:
            # what the REFERENCE's algorithm does for this frame (every triangle of every visited node's list tested, every child box slab-tested);
            # the kernel reaches the same pixels while skipping most of it (result-preserving index), so this is a work ratio, not a utilisation
            flops = algorithmic_flops(counters)
            alg_tflops = flops * per_rank / (kernel_ms * 1e-3) / 1e12
            roofline["reference_algorithm"] = {"flops_per_launch": flops * per_rank, "tflops_equivalent": round(alg_tflops, 3),
                                               "work_skipped_vs_reference": round(flops / prof["executed_f64_flop_per_launch_upper_bound"], 1) if prof and prof.get("executed_f64_flop_per_launch_upper_bound") else None,
                                               "equivalent_over_peak": round(alg_tflops / FP64_VECTOR_PEAK_TFLOPS, 3),
                                               "flop_model": "52*tri_tests + 24*aabb_tests + 250*hits_shaded (SURVEY.md 8d), counts from the oracle",
                                               "counters": {k: (int(v) if float(v).is_integer() else v) for k, v in counters.items() if k != "seconds_8_threads_container"},
                                               "counters_source": counters_src,
                                               "rays_all_kinds": int(counters["rays_primary"] + counters["rays_shadow"] + counters["rays_reflect"])}

        teapot = scene_name == "model2.obj"
        out = {"metric": "Mrays/s (primary) at 1920x1080, Utah teapot (model2.obj)" if (teapot and (W, H) == (1920, 1080)) else f"Mrays/s (primary) at {W}x{H}, {scene_name}",
               "value": round(rays_primary / (elapsed / args.steps) / 1e6, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
               "data": "reference scene assets/model2.obj (teapot+table+mirror, 6334 triangles), camera/lights of main.rs" if teapot else f"synthetic ({scene_name}: SURVEY.md 8d soup recipe)" if scene_name.startswith("soup") else f"reference scene {scene_name}",
               "config": {"workload": f"{scene_name} {W}x{H}, 4 sub-samples/pixel, shadow rays + depth-5 mirror reflection, f64",
                          "rays_primary_per_frame": rays_primary,
                          "partition": "single launch" if not multi else f"8x8-pixel tiles round-robin over {world} GPUs + RCCL gather to GPU 0, " + (f"inside the library (rrt_multi_enqueue), {DEPTH} frames in flight" if in_lib else "torch.distributed.gather, " + ("one frame at a time" if simple[0] else f"{DEPTH} frames in flight (one stream per slot)")),
                          "octree_nodes": rt.info["n_nodes"], "triangles": rt.info["n_tris"], "filter_variant": rrt.VARIANT_NAMES[rt.last_stats()["filter_variant"]]},
               "frame_ms": round(ms_per_step, 4), "kernel_ms": round(kernel_ms, 4), "host_enqueue_ms_per_step": round(host_enqueue_ms, 4), "setup_ms": setup,
               **({"rehearsal": "gloo on one GPU: NOT a benchmark result"} if rehearsal else {}),
               "frame_checksum": int(np.bitwise_xor.reduce(frame.ravel().astype(np.uint64) * np.arange(1, frame.size + 1, dtype=np.uint64)))}
        if multi:
            out.update({"gather": gather_label, "pipeline_fallback": pipeline_fallback, "pipeline_error": pipeline_error, "kernel_ms_per_rank_min": round(kernel_ms_min, 4),
                        "kernel_ms_per_rank_max": round(kernel_ms_max, 4), "gather_ms": round(gather_ms, 4) if gather_ms is not None else None})
        if host_fb is not None:
            out.update({"frame_ms_host_fb": host_fb["frame_ms_host_fb"], "host_fb": host_fb})
        if first is not None:
            out.update({"first_frame_ms": first["first_frame_ms"], "first_frame": first})
        out["roofline"] = roofline
        if cpu is not None:
            out["cpu_baseline"] = cpu
        return out

    def emit(out):
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)

    # ---- the measurement.  N > 1: this process is one rank of ONE gather path's child group (the GPU-free supervisor runs the paths one after the
    # other in fresh processes and merges their lines): it either finishes or exits non-zero -- no in-process watchdog, no forced zero exit with the GPU held.
    if not multi:
        res = measure(step, fence)
        if rank == 0: emit(build_out(res, None))
    elif args.gather == "torch" or rehearsal:
        stage("torch.distributed.gather path: timed steps")
        res = measure(torch_step, torch_fence)
        if rank == 0:
            out = build_out(res, "torch")
            out["frames_in_flight"] = 1 if simple[0] else DEPTH
            emit(out)
    else:
        # the library's own gather (rrt_dist_create / rrt_multi_enqueue): the cheaper host side, one C call per frame
        stage("library gather: binding RCCL inside the library (dlopen)")
        if os.environ.get("RRT_BENCH_TEST_HANG") == "lib":      # test hook (tests/test_gpu_parity.py): a library path that never finishes
            stage("library gather: test hook RRT_BENCH_TEST_HANG=lib, sleeping before rrt_dist_create")
            time.sleep(1e6)
        lib_mg = lib_init()
        if lib_mg is None:
            dist.destroy_process_group()
            raise SystemExit(f"bench.py: library gather could not be set up: {lib_state['error']}")
        stage("library gather: first frames through rrt_multi_enqueue")
        lib_enqueue = lib_mg.bind_enqueue(fb if rank == 0 else None, W, H)

        def lib_step(i):
            lib_enqueue()                                # trace -> grouped RCCL send/recv to rank 0 -> de-tile, all enqueued inside the library

        def lib_fence():
            set_stream(default_stream)
            lib_mg.sync()
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
        for _ in range(DEPTH): lib_step(None)
        lib_fence()
        stage("library gather: timed steps")
        res = measure(lib_step, lib_fence, lib=lib_mg)
        if rank == 0:
            out = build_out(res, "lib")
            out["frames_in_flight"] = DEPTH
            emit(out)

    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
