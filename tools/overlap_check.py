"""Developer tool: does tracing consecutive frames on alternating streams hide the launch tail?  One rank's share of an N-GPU frame (tile k -> rank k % N)
   is rendered `steps` times on ONE stream, then alternating over TWO streams.   python tools/overlap_check.py [world] [W H]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rrt = importlib.import_module("rust-ray-tracer_amd")
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
sd = rrt.parse_obj_file(os.path.join(ROOT, "assets/model2.obj"))
rt = rrt.RayTracer(sd, rrt.default_lights(), box_filter=os.environ.get("RRT_FILTER") or None)
tpr = rrt.tiles_per_rank(W, H, world)
steps = 400
for ns in (1, 2, 3, 4):
    bufs = [torch.zeros(tpr * 64, dtype=torch.int32, device="cuda") for _ in range(ns)]
    streams = [torch.cuda.Stream() for _ in range(ns)]
    launch = [rt.bind_render_tiles(bufs[j], W, H, 0, world, stream=streams[j].cuda_stream) for j in range(ns)]
    for i in range(20): launch[i % ns]()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps): launch[i % ns]()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"variant {rt.last_stats()['filter_variant']} world {world} {W}x{H} {ns} stream(s): {dt * 1e3:.4f} ms per rank-frame  (x{world} ranks -> {4 * (W // 2 * 2) * (H // 2 * 2 - 1) / dt / 1e6:.0f} Mrays/s if every rank kept this pace)", flush=True)
