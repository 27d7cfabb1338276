"""Developer tool: how long the waves of a frame live (needs the -DRRT_PROFILE -DRRT_PROF_WAVETIME build as librrt_hip_prof.so; tools/build_variant.sh prof "-DRRT_PROFILE -DRRT_PROF_WAVETIME").
   python tools/wave_time_hist.py [W H] [scene|soupN]"""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RRT_LIB", os.path.join(ROOT, "rust-ray-tracer_amd", "librrt_hip_wt.so"))
rrt = importlib.import_module("rust-ray-tracer_amd")
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
scene = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "assets/model2.obj")
if scene.startswith("soup"):
    syn = importlib.import_module("rust-ray-tracer_amd.synthetic"); n = int(scene[4:])
    scene = syn.ensure_soup(os.path.join(ROOT, "assets"), n, syn.SEED_100K if n == 100000 else syn.SEED_1M)
sd = rrt.parse_obj_file(scene)
rt = rrt.RayTracer(sd, rrt.default_lights(), box_filter=os.environ.get("RRT_FILTER") or None)
L = rrt.lib(); L.rrt_prof_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
buf = (C.c_uint64 * 24)()
rt.render(W, H); L.rrt_prof_counters(rt._h, buf)
rt.render(W, H); L.rrt_prof_counters(rt._h, buf)
c = list(buf); tot = sum(c[0:16]) or 1
print(f"{tot:,d} waves, kernel {rt.last_stats()['kernel_ms']:.3f} ms (instrumented build), longest wave {c[16] / 100.0:.1f} us")
acc = 0
for b in range(16):
    if not c[b]: continue
    acc += c[b]
    lo = 0 if b == 0 else 2 ** b
    print(f"  {lo:6d} .. {2 ** (b + 1):6d} us: {c[b]:>9,d} waves  {100.0 * c[b] / tot:6.2f} %   (cumulative {100.0 * acc / tot:6.2f} %)")
