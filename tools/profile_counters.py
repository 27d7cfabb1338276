"""Developer tool: wave-level work counters of the trace kernel (needs `make -C rust-ray-tracer_amd/csrc prof`).
   RRT_LIB=rust-ray-tracer_amd/librrt_hip_prof.so python tools/profile_counters.py [W H] [scene.obj]"""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RRT_LIB", os.path.join(ROOT, "rust-ray-tracer_amd", "librrt_hip_prof.so"))
rrt = importlib.import_module("rust-ray-tracer_amd")
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
scene = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "assets/model2.obj")
if scene.startswith("soup"):   # soup100000 / soup1000000: generated on the spot (the .obj files are not shipped)
    syn = importlib.import_module("rust-ray-tracer_amd.synthetic"); n = int(scene[4:])
    scene = syn.ensure_soup(os.path.join(ROOT, "assets"), n, syn.SEED_100K if n == 100000 else syn.SEED_1M if n == 1000000 else 0x5EED0003)
sd = rrt.parse_obj_file(scene)
rt = rrt.RayTracer(sd, rrt.default_lights(), box_filter=os.environ.get("RRT_FILTER") or None)
L = rrt.lib()
L.rrt_prof_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
buf = (C.c_uint64 * 24)()
rt.render(W, H); L.rrt_prof_counters(rt._h, buf)       # warm + clear
rt.render(W, H); ms = rt.last_stats()["kernel_ms"]; L.rrt_prof_counters(rt._h, buf)
c = list(buf)
names = ["node_visits(wave)", "node_visit_lanes", "tri_iters(wave)", "tri_lane_tests", "tri_box_tests(wave)", "single_candidate_decided_in_fp32(wave)", "traverse_calls(wave)", "traverse_lanes", "slab_iters(wave)", "slab_lane_tests", "super_tests(wave)", "cluster_tests(wave)", "internal_visits(wave)", "shade_blocks(wave-divergent)", "slab_exact_fallbacks(wave)"]
for i, n in enumerate(names):
    print(f"{n:24s} {c[i]:>16,d}")
print(f"kernel_ms (counters build) {ms:.2f}")
tn = ["pick node + record load", "children (reach, quotients, slab, rank)", "own list (boxes + MT)", "push / unwind", "shading + state machine", "traverse set-up"]
tt = sum(c[16:22]) or 1
for i, n in enumerate(tn):
    print(f"  time share {n:44s} {100.0 * c[16 + i] / tt:5.1f} %")
print(f"lane utilisation in triangle loop: {c[3] / max(1, 64 * c[2]):.3f}   in slab tests: {c[9] / max(1, 64 * c[8]):.3f}   at node visits: {c[1] / max(1, 64 * c[0]):.3f}   at traverse(): {c[7] / max(1, 64 * c[6]):.3f}")
print(f"wave-iterations per SIMD-second budget: tri {c[2]:,d} slab {c[8]:,d}; cycles/tri-iter if all time were the triangle loop: {ms*1e-3*2.4e9*1024/max(1,c[2]):.0f}")
