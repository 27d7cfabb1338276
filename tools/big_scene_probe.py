#!/usr/bin/env python3
"""Developer tool (run ON THE GPU BOX via gpurun): scenes well beyond BASELINE.json's largest (1 M triangles) -- do the set-up kernels, the index and the
trace kernels hold at 4 M / 16 M triangles (32-bit offsets, 26-bit node ids in the pick key, the pinned ring, group records)?

    python tools/big_scene_probe.py <n_triangles> [--width 3840 --height 2160] [--no-host]  ->  gpurun_out/big_scene_<n>.json

  * triangle soup as SURVEY.md 8d describes it (centres uniform in [-4.5,4.5]x[0.5,5.5]x[-4.5,4.5], half-size 0.05), generated with numpy;
  * rrt_raytracer_create_from_arrays (GPU set-up) against rrt_model_from_arrays + RRT_FLAG_HOST_SETUP: octree arrays and every scene buffer byte-identical;
  * one frame, timed; a 16-row band of it ray by ray (rrt_get_ray_colours) in the indexed mode and in the reference-order mode (RRT_FLAG_NO_CULL): identical.
The oracle is not used here (tests/test_gpu_big_scene.py holds the 4 M-triangle scene against it).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ORIGIN = (0.0, 2.0, -10.0)
MATS = [dict(ka=(1, 1, 1), kd=(1, 1, 1), ks=(1, 1, 1), ns=240.0, kr=0.0, tex=0, bump=-1)]
OCT_KEYS = ("aabb", "first_child", "tri_count", "own_off", "own_idx")
SCENE_BUFS = ("nodes", "geom", "attr", "supers", "cboxes", "child_boxes", "tboxes")


def soup(n, seed):
    rng = np.random.default_rng(seed)
    c = np.empty((n, 1, 3)); c[:, 0, 0] = rng.uniform(-4.5, 4.5, n); c[:, 0, 1] = rng.uniform(0.5, 5.5, n); c[:, 0, 2] = rng.uniform(-4.5, 4.5, n)
    pos = c + rng.uniform(-0.05, 0.05, (n, 3, 3))
    e1, e2 = pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0]
    fn = np.cross(e1, e2); fn /= np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-300)
    nrm = np.repeat(fn[:, None, :], 3, axis=1).copy()
    uv = np.zeros((n, 3, 3)); uv[:, :, :2] = rng.random((n, 3, 2))
    return np.ascontiguousarray(pos), uv, nrm, np.zeros(n, np.uint32)


def row_dirs(w, h, r, xs):      # engine.rs:207-236 + put_pixel's row flip (engine.rs:147-150)
    y = (h - h // 2) - r
    x = np.asarray(xs, np.float64) - (w // 2)
    d = np.empty((4, len(x), 3)); d[..., 2] = 1.0
    d[0, :, 0] = x * (1.0 / w); d[1, :, 0] = (x + 0.5) * (1.0 / w); d[2, :, 0] = d[0, :, 0]; d[3, :, 0] = d[1, :, 0]
    d[0, :, 1] = y * (1.0 / h); d[1, :, 1] = d[0, :, 1]; d[2, :, 1] = (y + 0.5) * (1.0 / h); d[3, :, 1] = d[2, :, 1]
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("n", type=int)
    ap.add_argument("--width", type=int, default=3840); ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--no-host", action="store_true")
    ap.add_argument("--band-rows", type=int, default=16); ap.add_argument("--frames", type=int, default=5)
    a = ap.parse_args()
    rrt = importlib.import_module("rust-ray-tracer_amd")
    out = {"n_triangles": a.n, "width": a.width, "height": a.height}

    def lap(msg, t0):
        dt = time.perf_counter() - t0; print(f"[big scene] {msg}: {dt * 1e3:.1f} ms", flush=True); return dt * 1e3

    t0 = time.perf_counter(); pos, uv, nrm, mat = soup(a.n, 0xB16 + a.n); lap("generated", t0)
    tex = [np.full((2, 2, 3), 200, np.uint8)]; lights = rrt.default_lights()
    rrt.RayTracer.from_arrays(pos[:64], uv[:64], nrm[:64], mat[:64], MATS, tex, lights)           # context, ring, code objects
    t0 = time.perf_counter(); gpu = rrt.RayTracer.from_arrays(pos, uv, nrm, mat, MATS, tex, lights); out["create_from_arrays_ms"] = lap("rrt_raytracer_create_from_arrays (GPU set-up)", t0)
    out["setup_times"] = gpu.setup_times(); tree = gpu.octree(); out["info"] = tree["info"]
    print("[big scene] info", tree["info"], "set-up", out["setup_times"], flush=True)

    if not a.no_host:
        t0 = time.perf_counter(); sd = rrt.SceneData.from_arrays(pos, uv, nrm, mat, MATS, tex); lap("rrt_model_from_arrays", t0)
        t0 = time.perf_counter(); host = rrt.RayTracer(sd, lights, host_setup=True); out["create_host_setup_ms"] = lap("rrt_raytracer_create (host set-up)", t0)
        ht = sd.octree()
        for k in OCT_KEYS:
            assert tree[k].shape == ht[k].shape and np.array_equal(tree[k].view(np.uint8), ht[k].view(np.uint8)), f"octree array {k} differs between the GPU and the host build"
        for name in SCENE_BUFS:
            g, h = gpu.buffer(name), host.buffer(name)
            assert g.shape == h.shape and np.array_equal(g, h), f"scene buffer {name} differs between the GPU and the host set-up ({g.shape} vs {h.shape})"
            del g, h
        out["gpu_build_equals_host_build"] = True; print("[big scene] GPU set-up == host set-up, byte for byte", flush=True)
        del host, ht

    w, h = a.width, a.height
    frame = gpu.render(w, h); ms = []
    for _ in range(a.frames):
        t0 = time.perf_counter(); f2 = gpu.render(w, h); ms.append((time.perf_counter() - t0) * 1e3)
        assert np.array_equal(f2, frame)
    st = gpu.last_stats(); out["frame_ms_host_fb_median"] = float(np.median(ms)); out["kernel_ms"] = st.get("kernel_ms"); out["walk"] = st.get("walk")
    out["covered"] = float(((frame != 0xFFFFFF) & (frame != 0)).mean())
    print(f"[big scene] frame {w}x{h}: {np.median(ms):.2f} ms into host memory, kernel {st.get('kernel_ms')}, covered {out['covered']:.3f}", flush=True)

    exact = rrt.RayTracer.from_arrays(pos, uv, nrm, mat, MATS, tex, lights, no_cull=True)
    r0 = h // 2 - a.band_rows // 2; xs = np.arange(0, w)
    d = np.concatenate([row_dirs(w, h, r, xs).reshape(-1, 3) for r in range(r0, r0 + a.band_rows)]); o = np.tile(ORIGIN, (len(d), 1))
    t0 = time.perf_counter(); c_fast = gpu.get_ray_colours(o, d); lap("band, indexed mode", t0)
    t0 = time.perf_counter(); c_exact = exact.get_ray_colours(o, d); lap("band, reference-order mode", t0)
    assert np.array_equal(c_fast, c_exact), f"{(c_fast != c_exact).sum()} of {len(d)} band rays differ between the indexed and the reference-order mode"
    ch = lambda c: np.stack([(c >> 16) & 255, (c >> 8) & 255, c & 255], -1).astype(np.int64)
    mixed = ch(c_fast.reshape(a.band_rows, 4, w)).sum(1) // 4
    assert np.array_equal(mixed, ch(frame[r0:r0 + a.band_rows])), "band rays mixed per pixel differ from the frame's rows"
    out["band_rays_identical"] = int(len(d)); print(f"[big scene] {len(d)} band rays identical in both modes and equal to the frame's rows", flush=True)

    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"big_scene_{a.n}.json"), "w"), indent=1, default=str)
    print(json.dumps(out, default=str))


if __name__ == "__main__":
    main()
