"""Developer experiment: how much does sorting shadow rays by the triangle they start on buy?  Primary rays of a frame -> (t, triangle) through
rrt_intersect_rays; the shadow rays of light 1 built on the host; timed (rocprofv3 --kernel-trace --stats sees the two intersect_kernel launches)
in pixel order and sorted by triangle.   python tools/shadow_sort_probe.py soup100000 1920 1080"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rrt = importlib.import_module("rust-ray-tracer_amd"); syn = importlib.import_module("rust-ray-tracer_amd.synthetic")
scene, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
if scene.startswith("soup"):
    n = int(scene[4:]); scene = syn.ensure_soup(os.path.join(ROOT, "assets"), n, syn.SEED_100K if n == 100000 else syn.SEED_1M)
sd = rrt.parse_obj_file(scene)
rt = rrt.RayTracer(sd, rrt.default_lights(), box_filter=os.environ.get("RRT_FILTER", "lane"))
# primary rays in the kernel's wave order: 4x4-pixel blocks x 4 sub-samples
ys, xs = np.meshgrid(np.arange(1, H), np.arange(0, W), indexing="ij")
def wave_order(a):   # [H-1, W] -> blocks of 4x4
    hh = (a.shape[0] // 4) * 4; ww = (a.shape[1] // 4) * 4
    return a[:hh, :ww].reshape(hh // 4, 4, ww // 4, 4).transpose(0, 2, 1, 3).reshape(-1)
px = wave_order(xs); py = wave_order(ys)
sub = np.tile(np.arange(4), len(px)); px = np.repeat(px, 4); py = np.repeat(py, 4)
x = px - W // 2; y = (H - H // 2) - py
d = np.stack([(x + 0.5 * (sub & 1)) * (1.0 / W), (y + 0.5 * (sub >> 1)) * (1.0 / H), np.ones(len(x))], -1)
o = np.tile([0.0, 2.0, -10.0], (len(d), 1))
t0 = time.time(); hit, t, u, v, tri = rt.intersect_rays(o, d); print("primary", len(d), "rays", round(time.time() - t0, 2), "s  hit", hit.mean(), flush=True)
p = o[hit] + d[hit] * t[hit, None]
nrm = -d[hit] / np.linalg.norm(d[hit], axis=1, keepdims=True)
L = np.array([-7.0, 1.0, -15.0])
so = p + nrm * 1e-4; sdir = L - p; smax = np.linalg.norm(sdir, axis=1)
t0 = time.time(); a = rt.intersect_rays(so, sdir, smax); print("shadow, pixel order", round(time.time() - t0, 2), "s occluded", a[0].mean(), flush=True)
order = np.argsort(tri[hit], kind="stable")
t0 = time.time(); b = rt.intersect_rays(so[order], sdir[order], smax[order]); print("shadow, sorted by triangle", round(time.time() - t0, 2), "s", flush=True)
assert np.array_equal(a[0][order], b[0])
