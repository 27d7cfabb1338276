"""Ad-hoc GPU sanity run: HIP path vs oracle on the teapot at small sizes, then a 1080p timing.  (Not a test; see tests/.)"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
rrt = importlib.import_module("rust-ray-tracer_amd")
from oracle import binding as ob

sd = rrt.parse_obj_file(os.path.join(ROOT, "assets/model2.obj"))
print(sd.info, flush=True)
pos, uv, nrm, mat = sd.triangles()
lights = rrt.default_lights()
osc = ob.OracleScene(pos, uv, nrm, mat, sd.materials(), sd.textures(), [(l.kind, l.intensity, (l.v.x, l.v.y, l.v.z)) for l in lights], (0, 2, -10))
rt = rrt.RayTracer(sd, lights)
rt_exact = rrt.RayTracer(sd, lights, no_cull=True)
for (w, h) in [(64, 48), (160, 120), (321, 241), (640, 480)]:
    t0 = time.time(); g = rt.render(w, h); tg = time.time() - t0
    t0 = time.time(); o, cnt = osc.render(w, h); to = time.time() - t0
    ch = lambda a: np.stack([(a >> 16) & 255, (a >> 8) & 255, a & 255], -1).astype(int)
    d = np.abs(ch(g) - ch(o))
    print(f"{w}x{h}: gpu {tg*1e3:.1f} ms (kernel {rt.last_stats()['kernel_ms']:.2f} ms) oracle {to:.2f} s  maxdiff {d.max()}  n_diff_px {(d.max(-1) > 0).sum()}  n_gt1 {(d.max(-1) > 1).sum()}", flush=True)
for (w, h) in [(640, 480), (1920, 1080)]:
    a = rt.render(w, h); ta = rt.last_stats()['kernel_ms']; b = rt_exact.render(w, h); tb = rt_exact.last_stats()['kernel_ms']
    print(f"{w}x{h}: cull {ta:.2f} ms  no_cull {tb:.2f} ms  identical {np.array_equal(a, b)}  n_diff {(a != b).sum()}", flush=True)
for rep in range(3):
    t0 = time.time(); g = rt.render(1920, 1080); tg = time.time() - t0
    st = rt.last_stats()
    print(f"1080p: wall {tg*1e3:.1f} ms kernel {st['kernel_ms']:.2f} ms  -> {st['rays_primary']/st['kernel_ms']/1e3:.1f} Mrays/s primary", flush=True)
