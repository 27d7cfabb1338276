"""Developer tool: kernel time of every BASELINE.json config that fits one GPU (single launch), plus the filter variant picked and a
bit-compare against the reference-order mode (RRT_FLAG_NO_CULL) where that is affordable."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rrt = importlib.import_module("rust-ray-tracer_amd"); syn = importlib.import_module("rust-ray-tracer_amd.synthetic")
A = os.path.join(ROOT, "assets")
rows = []
def run(name, path, w, h, compare=True):
    sd = rrt.parse_obj_file(path)
    rt = rrt.RayTracer(sd, rrt.default_lights())
    fb = rt.render(w, h)
    ts = []
    for _ in range(5):
        fb = rt.render(w, h); ts.append(rt.last_stats()["kernel_ms"])
    st = rt.last_stats()
    same = None
    if compare:
        ex = rrt.RayTracer(sd, rrt.default_lights(), no_cull=True)
        same = bool(np.array_equal(ex.render(w, h), fb)); t_ex = ex.last_stats()["kernel_ms"]
    row = dict(config=name, tris=sd.info["n_tris"], nodes=sd.info["n_nodes"], size=f"{w}x{h}", kernel_ms=round(float(np.median(ts)), 3),
               mrays_primary=round(st["rays_primary"] / np.median(ts) / 1e3, 1), filter=rrt.VARIANT_NAMES[st["filter_variant"]],
               identical_to_no_cull=same, no_cull_ms=round(t_ex, 2) if compare else None)
    rows.append(row); print(json.dumps(row), flush=True)
run("teapot 640x480 (configs[0] scene)", os.path.join(A, "model2.obj"), 640, 480)
run("teapot 1920x1080 (configs[1], headline)", os.path.join(A, "model2.obj"), 1920, 1080)
run("teapot 3840x2160 (configs[3] on one GPU)", os.path.join(A, "model2.obj"), 3840, 2160)
run("model3.obj 1920x1080", os.path.join(A, "model3.obj"), 1920, 1080)
run("100k soup 1920x1080 (configs[2])", syn.ensure_soup(A, 100000, syn.SEED_100K), 1920, 1080)
run("1M soup 3840x2160 (configs[4] on one GPU)", syn.ensure_soup(A, 1000000, syn.SEED_1M), 3840, 2160, compare=False)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "config_table.json"), "w"), indent=1)
