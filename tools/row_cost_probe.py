"""Developer tool: where in the frame the time goes -- rrt_render_progressive in bands of `rows` canvas rows, wall time per band (kernel + synchronisation).
   python tools/row_cost_probe.py [rows=40] [walk=bundle]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rrt = importlib.import_module("rust-ray-tracer_amd")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 40
walk = sys.argv[2] if len(sys.argv) > 2 else "bundle"
sd = rrt.parse_obj_file(os.path.join(ROOT, "assets", "model2.obj"))
rt = rrt.RayTracer(sd, rrt.default_lights(), box_filter=walk)
W, H = 1920, 1080
rt.render(W, H)
for rep in range(3):
    stamps = []
    t0 = time.perf_counter()
    rt.render_progressive(W, H, lambda fb, first_row, n_rows: stamps.append((first_row, n_rows, time.perf_counter())), chunk_rows=rows)
    if rep == 2:
        prev = t0
        for first_row, n_rows, t in stamps:
            print(f"rows {first_row:4d}..{first_row + n_rows - 1:4d}: {1e3 * (t - prev):6.3f} ms"); prev = t
        print(f"total {1e3 * (prev - t0):.3f} ms in {len(stamps)} bands")
