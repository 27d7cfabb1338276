"""Maximum sizes: a scene four times BASELINE.json's largest (4 M triangles; RRT_BIG_SCENE_N overrides), straight from the caller's arrays.

  * the octree built on the GPU (csrc/scene_build.hip) == the host build (csrc/octree.cpp) == the oracle's one-at-a-time insertion (octree.rs:41-241),
    byte for byte; every buffer the trace kernels read == what the host set-up uploads;
  * a 3840x2160 frame; a band of it ray by ray in the indexed and in the reference-order mode (RRT_FLAG_NO_CULL): identical, and equal to the
    frame's rows after Color::mix (entities.rs:49-69);
  * a sample of the band's rays within +-1 per channel of the oracle (measured 0).
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import channels, lights_tuple
from test_gpu_build import MATS, OCT_KEYS, SCENE_BUFS, TEX, assert_same_octree
from test_gpu_configs import ORIGIN, row_dirs

pytestmark = pytest.mark.gpu
N = int(os.environ.get("RRT_BIG_SCENE_N", 4_000_000))


def soup(n, seed):
    """SURVEY.md 8d's soup (centres uniform in [-4.5,4.5]x[0.5,5.5]x[-4.5,4.5], vertices within 0.05 of them), vectorised."""
    rng = np.random.default_rng(seed)
    c = np.empty((n, 1, 3)); c[:, 0, 0] = rng.uniform(-4.5, 4.5, n); c[:, 0, 1] = rng.uniform(0.5, 5.5, n); c[:, 0, 2] = rng.uniform(-4.5, 4.5, n)
    pos = np.ascontiguousarray(c + rng.uniform(-0.05, 0.05, (n, 3, 3)))
    fn = np.cross(pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0]); fn /= np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-300)
    uv = np.zeros((n, 3, 3)); uv[:, :, :2] = rng.random((n, 3, 2))
    return pos, uv, np.repeat(fn[:, None, :], 3, axis=1).copy(), np.zeros(n, np.uint32)


def test_four_million_triangles_from_arrays(rrt, ob):
    pos, uv, nrm, mat = soup(N, 0xB16)
    lights = rrt.default_lights()
    gpu = rrt.RayTracer.from_arrays(pos, uv, nrm, mat, MATS, TEX, lights)
    tree = gpu.octree(); info = tree["info"]
    assert info["n_tris"] == N and info["n_nodes"] % 8 == 1
    t = gpu.setup_times()
    print(f"{N} triangles: {info['n_nodes']} nodes, depth {info['max_depth']}, root list {info['root_own_count']}; GPU set-up: octree {t['octree_ms']:.1f} ms, index {t['index_ms']:.1f} ms, create {t['create_ms']:.1f} ms")

    sd = rrt.SceneData.from_arrays(pos, uv, nrm, mat, MATS, TEX)
    host = rrt.RayTracer(sd, lights, host_setup=True)
    assert_same_octree(tree, sd.octree(), "GPU build vs host build")
    for name in SCENE_BUFS:
        g, h = gpu.buffer(name), host.buffer(name)
        assert g.shape == h.shape and np.array_equal(g, h), f"scene buffer {name} differs between the GPU and the host set-up"
    del host, g, h
    osc = ob.OracleScene(pos, uv, nrm, mat, MATS, TEX, lights_tuple(lights), ORIGIN)
    assert_same_octree(tree, osc.octree(), "GPU build vs oracle build")

    w, h = 3840, 2160
    frame = gpu.render(w, h)
    assert np.array_equal(frame[0], np.zeros(w, np.uint32)) and ((frame != 0xFFFFFF) & (frame != 0)).mean() > 0.5
    rows = 8; r0 = h // 2 - rows // 2; xs = np.arange(0, w)
    d = np.concatenate([row_dirs(w, h, r, xs).reshape(-1, 3) for r in range(r0, r0 + rows)]); o = np.tile(ORIGIN, (len(d), 1))
    c_fast = gpu.get_ray_colours(o, d)
    c_exact = rrt.RayTracer.from_arrays(pos, uv, nrm, mat, MATS, TEX, lights, no_cull=True).get_ray_colours(o, d)
    assert np.array_equal(c_fast, c_exact), f"{(c_fast != c_exact).sum()} of {len(d)} band rays differ between the indexed and the reference-order mode"
    assert np.array_equal(channels(c_fast.reshape(rows, 4, w)).sum(1) // 4, channels(frame[r0:r0 + rows])), "band rays mixed per pixel differ from the frame's rows"
    idx = np.random.default_rng(6).choice(len(d), 1500, replace=False)
    with ThreadPoolExecutor(16) as pool:
        want = np.fromiter(pool.map(lambda i: osc.get_ray_colour(ORIGIN, d[i]), idx), np.uint32, len(idx))
    diff = np.abs(channels(want) - channels(c_fast[idx])).max()
    assert diff <= 1, f"sampled band rays differ from the oracle by {diff}"
    print(f"frame kernel {gpu.last_stats()['kernel_ms']:.2f} ms; {len(d)} band rays identical in both modes; {len(idx)} sampled rays within {diff} of the oracle")
