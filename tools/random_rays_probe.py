"""Developer tool: time rrt_intersect_rays on scattered (random) rays for each traversal variant.  python tools/random_rays_probe.py [scene] [n]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rrt = importlib.import_module("rust-ray-tracer_amd")
scene = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "assets/model2.obj")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
if scene.startswith("soup"):
    syn = importlib.import_module("rust-ray-tracer_amd.synthetic"); k = int(scene[4:])
    scene = syn.ensure_soup(os.path.join(ROOT, "assets"), k, syn.SEED_100K if k == 100000 else syn.SEED_1M if k == 1000000 else 0x5EED0003)
sd = rrt.parse_obj_file(scene)
rng = np.random.default_rng(5)
o = rng.uniform([-5, 0, -8], [5, 6, 5], (n, 3)); d = rng.normal(size=(n, 3))
ref = None
for mode in ("lane", "bundle", "ray", None):
    rt = rrt.RayTracer(sd, rrt.default_lights(), box_filter=mode)
    rt.intersect_rays(o[:4096], d[:4096])
    t0 = time.perf_counter(); r = rt.intersect_rays(o, d); dt = time.perf_counter() - t0
    same = True if ref is None else all(np.array_equal(a, b) for a, b in zip(r, ref))
    ref = ref or r
    st = rt.last_stats()
    print(f"{str(mode):7s} {n} random rays: kernel {st['kernel_ms']:8.2f} ms ({n / st['kernel_ms'] / 1e3:8.1f} Mrays/s), wall {dt * 1e3:8.1f} ms incl. transfers; variant used: {rrt.VARIANT_NAMES[st['filter_variant']]}; identical to first: {same}; hits {int(r[0].sum())}")
