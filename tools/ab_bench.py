"""Developer tool: interleaved A/B timing of two builds of librrt_hip.so in ONE process on ONE device (cdna guide rule 24).
   python tools/ab_bench.py path/to/libA.so path/to/libB.so [W H] [rounds]"""
import ctypes as C, importlib, importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
libs = sys.argv[1:3]
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 15
scene = os.path.join(ROOT, "assets/model2.obj")
mods = []
for i, path in enumerate(libs):
    os.environ["RRT_LIB"] = os.path.abspath(path)
    spec = importlib.util.spec_from_file_location(f"rrt_{i}", os.path.join(ROOT, "rust-ray-tracer_amd", "__init__.py"))
    m = importlib.util.module_from_spec(spec); sys.modules[f"rrt_{i}"] = m; spec.loader.exec_module(m)
    m.lib()
    mods.append(m)
rts = []
for m in mods:
    sd = m.parse_obj_file(scene)
    rts.append((m, sd, m.RayTracer(sd, m.default_lights())))
    print(os.path.basename(path), "filter_variant (autotuned):", rts[-1][2].last_stats().get("filter_variant"))
frames = [rt.render(W, H) for _, _, rt in rts]
print("frames identical:", all(np.array_equal(frames[0], f) for f in frames[1:]))
times = [[] for _ in rts]
for r in range(rounds):
    for i, (_, _, rt) in enumerate(rts):
        rt.render(W, H); times[i].append(rt.last_stats()["kernel_ms"])
for path, t in zip(libs, times):
    t = np.array(t)
    print(f"{os.path.basename(path):32s} median {np.median(t):.3f} ms  min {t.min():.3f}  max {t.max():.3f}")
