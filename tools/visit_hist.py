"""Developer tool: histogram of lanes parked at the visited node per wave-visit (needs the -DRRT_PROFILE -DRRT_PROF_HIST build as librrt_hip_prof.so).
   python tools/visit_hist.py [W H] [scene|soupN]"""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RRT_LIB", os.path.join(ROOT, "rust-ray-tracer_amd", "librrt_hip_prof.so"))
rrt = importlib.import_module("rust-ray-tracer_amd")
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
scene = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "assets/model2.obj")
if scene.startswith("soup"):
    syn = importlib.import_module("rust-ray-tracer_amd.synthetic"); n = int(scene[4:])
    scene = syn.ensure_soup(os.path.join(ROOT, "assets"), n, syn.SEED_100K if n == 100000 else syn.SEED_1M if n == 1000000 else 0x5EED0003)
sd = rrt.parse_obj_file(scene)
rt = rrt.RayTracer(sd, rrt.default_lights(), box_filter=os.environ.get("RRT_FILTER") or None)
L = rrt.lib(); L.rrt_prof_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
buf = (C.c_uint64 * 24)()
rt.render(W, H); L.rrt_prof_counters(rt._h, buf)
rt.render(W, H); L.rrt_prof_counters(rt._h, buf)
c = list(buf); tot = sum(c[0:6]) or 1
for name, v in zip(["1 lane", "2-3", "4-7", "8-15", "16-31", "32-64"], c[0:6]):
    print(f"visits with {name:6s} lanes: {v:>12,d}  {100.0 * v / tot:5.1f} %")
print(f"internal {c[6]:,d}  leaf {c[7]:,d}  lane-visits {c[8]:,d}  mean lanes/visit {c[8] / tot:.1f}")
