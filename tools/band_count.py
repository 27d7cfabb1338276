"""Developer tool: how many (ray, triangle) pairs of the BASELINE frames lie inside the index filter's (alpha, delta) band AND are accepted by the
reference's Moller-Trumbore -- the only pairs the default (indexed) mode could treat differently from the reference-order mode.
Needs `make -C rust-ray-tracer_amd/csrc band` (run here, ships as devbin/librrt_hip_band.bin) and a GPU.  Every frame is rendered with
RRT_FLAG_NO_CULL, so every triangle of every visited list is tested and counted.

    RRT_LIB=devbin/librrt_hip_band.bin python tools/band_count.py [out.json]"""
import ctypes as C, importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rrt = importlib.import_module("rust-ray-tracer_amd")
syn = importlib.import_module("rust-ray-tracer_amd.synthetic")
L = rrt.lib()
L.rrt_prof_band_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
A = os.path.join(ROOT, "assets")
configs = [("configs[0] teapot 640x480", os.path.join(A, "model2.obj"), 640, 480), ("configs[1] teapot 1920x1080", os.path.join(A, "model2.obj"), 1920, 1080),
           ("configs[2] 100k soup 1920x1080", syn.ensure_soup(A, 100000, syn.SEED_100K), 1920, 1080), ("configs[3] teapot 3840x2160", os.path.join(A, "model2.obj"), 3840, 2160),
           ("configs[4] 1M soup 3840x2160", syn.ensure_soup(A, 1000000, syn.SEED_1M), 3840, 2160)]
# ---- self-check of the counter: the construction of tests/test_gpu_configs.py case C -- 400 triangles built in f64 INSIDE planes through an apex that is
# not the raytracer's origin, 200 000 rays from that apex lying in those planes to rounding.  The reference's Moller-Trumbore accepts a few per cent of
# such pairs (rounding noise): every one of them is a secondary-ray pair (origin != the raytracer's) and must be counted as in-band.
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_configs as tc
rng = np.random.default_rng(55)
apex = np.array([1.5, 1.0, -8.0])
tris, planes = tc._plane_scene(rng, apex, 40, 10, 1000)
sd = tc._scene_data(rrt, tris)
O, D = tc._coplanar_rays(rng, tris, planes, 400, apex, 200_000)
rt = rrt.RayTracer(sd, rrt.default_lights(), no_cull=True)
buf = (C.c_uint64 * 4)()
L.rrt_prof_band_counters(rt._h, buf); hit = rt.intersect_rays(O, D)[0]; L.rrt_prof_band_counters(rt._h, buf)
print("self-check (200 000 in-plane rays from a foreign apex):", list(buf), "rays that hit:", int(hit.sum()), flush=True)
assert buf[1] > 0 and buf[2] == 0, "the counter did not see the constructed in-band pairs"
self_check = {"secondary_pairs_accepted": buf[0], "secondary_pairs_accepted_in_band": buf[1]}
del rt, sd

out = {"self-check: tests/test_gpu_configs.py case C (constructed in-plane rays from a foreign apex)": self_check}
for name, path, w, h in configs:
    sd = rrt.parse_obj_file(path)
    rt = rrt.RayTracer(sd, rrt.default_lights(), no_cull=True)
    buf = (C.c_uint64 * 4)()
    L.rrt_prof_band_counters(rt._h, buf)
    rt.render(w, h)
    L.rrt_prof_band_counters(rt._h, buf)
    out[name] = {"secondary_pairs_accepted": buf[0], "secondary_pairs_accepted_in_band": buf[1], "primary_pairs_accepted": buf[2], "primary_pairs_accepted_in_band": buf[3],
                 "kernel_ms_no_cull_counting_build": round(rt.last_stats()["kernel_ms"], 1)}
    print(name, out[name], flush=True)
    del rt, sd
if len(sys.argv) > 1:
    json.dump({"what": "pairs (ray, listed triangle) ACCEPTED by Moller-Trumbore (ray.rs:56-94) in the reference-order walk, and those of them inside the (alpha, delta) band "
                       "of DESIGN.md section 4 (direction within alpha of the triangle's plane and origin within delta of it): only these could be dropped by the fp32 box filters",
               "frames": out}, open(sys.argv[1], "w"), indent=1)
