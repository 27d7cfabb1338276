"""Developer tool: K frames back to back on ONE stream against the same K frames dealt over S streams (each with a framebuffer of its own): does the head of
frame k + 1 fill the thinly occupied tail of frame k?   python tools/overlap_frames_probe.py [scene] [w h]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rrt = importlib.import_module("rust-ray-tracer_amd")
scene = sys.argv[1] if len(sys.argv) > 1 else "model2.obj"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
sd = rrt.parse_obj_file(os.path.join(ROOT, "assets", scene))
rt = rrt.RayTracer(sd, rrt.default_lights())
ref = rt.render(W, H); rt.render(W, H); rt.render(W, H)          # (second frame of a size: the variants are measured, the fastest kept)
K = 60
for S in (1, 2, 3, 4, 1, 2):
    streams = [torch.cuda.Stream() for _ in range(S)]
    fbs = [torch.zeros((H, W), dtype=torch.int32, device="cuda") for _ in range(S)]
    launch = [rt.bind_render(fbs[i], W, H, streams[i].cuda_stream) for i in range(S)]
    for i in range(2 * S): launch[i % S]()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K): launch[k % S]()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K * 1e3
    same = all(bool((fb.cpu().numpy().view("uint32") == ref).all()) for fb in fbs)
    print(f"{S} stream(s): {dt:.4f} ms per frame, frames identical: {same}", flush=True)
