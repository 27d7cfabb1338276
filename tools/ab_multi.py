"""Developer tool: interleaved timing of several builds of librrt_hip.so in ONE process on ONE device.
   python tools/ab_multi.py [WxH] [scene|soupN] lib1.so lib2.so ...      (timing only: ablation builds render wrong frames)"""
import importlib, importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = sys.argv[1:]
W, H = 1920, 1080
if args and "x" in args[0] and args[0][0].isdigit(): W, H = map(int, args.pop(0).split("x"))
scene = os.path.join(ROOT, "assets/model2.obj")
if args and not args[0].endswith(".so"): scene = args.pop(0)
rts = []
for i, path in enumerate(args):
    os.environ["RRT_LIB"] = os.path.abspath(path)
    spec = importlib.util.spec_from_file_location(f"rrt_{i}", os.path.join(ROOT, "rust-ray-tracer_amd", "__init__.py"))
    m = importlib.util.module_from_spec(spec); sys.modules[f"rrt_{i}"] = m; spec.loader.exec_module(m); m.lib()
    sc = scene
    if scene.startswith("soup"):
        syn = importlib.import_module("rust-ray-tracer_amd.synthetic"); n = int(scene[4:])
        sc = syn.ensure_soup(os.path.join(ROOT, "assets"), n, syn.SEED_100K if n == 100000 else syn.SEED_1M if n == 1000000 else 0x5EED0003)
    sd = m.parse_obj_file(sc)
    rts.append(m.RayTracer(sd, m.default_lights(), box_filter=os.environ.get("RRT_FILTER") or None))
frames = [rt.render(W, H) for rt in rts]
for rt in rts: rt.render(W, H)
times = [[] for _ in rts]
for r in range(int(os.environ.get("ROUNDS", "15"))):
    for i, rt in enumerate(rts):
        rt.render(W, H); times[i].append(rt.last_stats()["kernel_ms"])
for path, t, f, rt in zip(args, times, frames, rts):
    t = np.array(t)
    print(f"{os.path.basename(path):36s} median {np.median(t):.3f} ms  min {t.min():.3f}  same_frame_as_first {bool(np.array_equal(f, frames[0]))} variant {rt.last_stats().get('filter_variant')}")
