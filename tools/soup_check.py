"""Ad-hoc: 100k-triangle soup (BASELINE config 3) at 1080p: default (cluster index) vs RRT_FLAG_NO_CULL, identical?  timing."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rrt = importlib.import_module("rust-ray-tracer_amd"); syn = importlib.import_module("rust-ray-tracer_amd.synthetic")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
seed = syn.SEED_100K if n == 100000 else syn.SEED_1M if n == 1000000 else 0x5EED0003
t0 = time.time(); path = syn.ensure_soup(os.path.join(ROOT, "assets"), n, seed); print("gen", round(time.time() - t0, 2), flush=True)
t0 = time.time(); sd = rrt.parse_obj_file(path); print("load+octree", round(time.time() - t0, 2), sd.info, flush=True)
t0 = time.time(); rt = rrt.RayTracer(sd, rrt.default_lights()); print("upload(cull)", round(time.time() - t0, 2), flush=True)
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
for _ in range(3):
    a = rt.render(W, H); print("cull kernel ms", round(rt.last_stats()["kernel_ms"], 2), flush=True)
if "--exact" in sys.argv:
    ex = rrt.RayTracer(sd, rrt.default_lights(), no_cull=True)
    b = ex.render(W, H); print("no_cull kernel ms", round(ex.last_stats()["kernel_ms"], 2), "identical", np.array_equal(a, b), flush=True)
print("non-white fraction", float(((a != 0xFFFFFF) & (a != 0)).mean()))
