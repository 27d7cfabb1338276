"""BASELINE.json configs[2], [3], [4] at FULL size on one MI355X, through the C ABI, against the CPU oracle.

    configs[2]  100 k-triangle soup (seed 0x5EED0001) @1920x1080
    configs[3]  model2.obj (teapot) @3840x2160
    configs[4]  1 M-triangle soup (seed 0x5EED0002) @3840x2160

A full oracle frame of these takes minutes to hours on the host, so each config is checked by
  (a) determinism and the frame invariants of engine.rs:146-158 (row 0 unwritten, nothing else black),
  (b) sampled frame rows x every k-th pixel against oracle.get_ray_colour of the pixel's four sub-sample rays + Color::mix (+-1 per channel:
      only pow(), raytracer.rs:295, may differ by an ulp between glibc and OCML),
  (c) >= 5000 sampled primary AND secondary (shadow-shaped, reflection-shaped) rays against oracle.intersect, BIT-EXACT (hit, triangle, t, u, v),
  (d) the default (indexed) mode against RRT_FLAG_NO_CULL (every own list walked in full, in list order, as ray.rs:119-129): the whole frame for the
      100 k soup and the teapot, the primary + secondary rays of a 64-row band (through rrt_get_ray_colours) for the 1 M soup.
The oracle calls run on a thread pool (ctypes releases the GIL).
"""
import importlib
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import ASSETS, channels, oracle_scene_for

pytestmark = pytest.mark.gpu
COLOUR_TOL = 1
ORIGIN = (0.0, 2.0, -10.0)
POOL = ThreadPoolExecutor(max(4, min(32, (os.cpu_count() or 4))))


def row_dirs(w, h, r, xs):
    """Directions of the 4 sub-sample rays (engine.rs:207-236) of the pixels `xs` (canvas columns) of canvas row r: [4, len(xs), 3]."""
    y = (h - h // 2) - r                                    # put_pixel: new_y = h - (y + h/2), engine.rs:147-150
    x = np.asarray(xs, np.float64) - (w // 2)
    d = np.empty((4, len(x), 3)); d[..., 2] = 1.0
    d[0, :, 0] = x * (1.0 / w); d[1, :, 0] = (x + 0.5) * (1.0 / w); d[2, :, 0] = d[0, :, 0]; d[3, :, 0] = d[1, :, 0]
    d[0, :, 1] = y * (1.0 / h); d[1, :, 1] = d[0, :, 1]; d[2, :, 1] = (y + 0.5) * (1.0 / h); d[3, :, 1] = d[2, :, 1]
    return d


def check_rows_against_oracle(frame, osc, w, h, rows, step):
    xs = np.arange(0, 2 * (w // 2), step)
    for r in rows:
        d = row_dirs(w, h, r, xs).reshape(-1, 3)
        cols = np.fromiter(POOL.map(lambda v: osc.get_ray_colour(ORIGIN, v), d), np.uint32, len(d)).reshape(4, len(xs))
        mixed = channels(cols).sum(0) // 4                   # Color::mix, entities.rs:49-69
        got = channels(frame[r, xs])
        bad = np.abs(got - mixed).max(-1) > COLOUR_TOL
        assert not bad.any(), f"row {r}: {bad.sum()} of {len(xs)} sampled pixels differ from the oracle by more than {COLOUR_TOL}"


def sample_rays(osc, w, h, n_primary, rng, lights):
    """n_primary random sub-sample rays of the frame + for each one that hits: the shadow-shaped rays to the point lights (origin on the surface,
    un-normalised direction, max_t = |dir|: raytracer.rs:164-188) and one reflection-shaped ray (raytracer.rs:79-82) about a perturbed normal."""
    rows = rng.integers(1, h, n_primary); cols = rng.integers(0, 2 * (w // 2), n_primary); sub = rng.integers(0, 4, n_primary)
    d = np.stack([row_dirs(w, h, r, [c])[s, 0] for r, c, s in zip(rows, cols, sub)])
    o = np.tile(ORIGIN, (n_primary, 1))
    prim = list(POOL.map(lambda i: osc.intersect(o[i], d[i]), range(n_primary)))
    so, sdir, smax = [], [], []
    for i, (hit, t, u, v, tri) in enumerate(prim):
        if not hit:
            continue
        p = o[i] + d[i] * t
        n = -d[i] / np.linalg.norm(d[i]) + rng.normal(size=3) * 0.3
        n /= np.linalg.norm(n)
        for l in lights:
            if l.kind == 1:
                dirv = np.array([l.v.x, l.v.y, l.v.z]) - p
                so.append(p + n * 1e-4); sdir.append(dirv); smax.append(np.linalg.norm(dirv))
        rd = d[i] - n * 2.0 * np.dot(d[i], n)
        so.append(p + n * 1e-4); sdir.append(rd / np.linalg.norm(rd)); smax.append(np.inf)
    O = np.concatenate([o, np.array(so).reshape(-1, 3)]); D = np.concatenate([d, np.array(sdir).reshape(-1, 3)])
    M = np.concatenate([np.full(n_primary, np.inf), np.array(smax)])
    return O, D, M


def check_rays_bit_exact(rt, osc, O, D, M, min_rays, min_hit_frac=0.05):
    assert len(O) >= min_rays, len(O)
    hit, t, u, v, tri = rt.intersect_rays(O, D, M)
    ref = list(POOL.map(lambda i: osc.intersect(O[i], D[i], M[i]), range(len(O))))
    n_hit = 0
    for i, (rh, rt_, ru, rv, rtri) in enumerate(ref):
        assert bool(hit[i]) == rh, f"ray {i}: hit {bool(hit[i])} vs oracle {rh}"
        if rh:
            n_hit += 1
            assert (t[i], u[i], v[i], tri[i]) == (rt_, ru, rv, rtri), f"ray {i}: ({t[i]!r}, {u[i]!r}, {v[i]!r}, {tri[i]}) vs oracle ({rt_!r}, {ru!r}, {rv!r}, {rtri})"
    assert n_hit >= min_hit_frac * len(O), (n_hit, len(O))
    return n_hit


def frame_invariants(rt, w, h):
    a = rt.render(w, h); b = rt.render(w, h)
    assert np.array_equal(a, b), "two renders of the same frame differ"
    assert (a[0] == 0).all(), "row 0 must stay unwritten (engine.rs:146-158)"
    assert (a[1:, : 2 * (w // 2)] != 0).all(), "a traced pixel is black: ambient light alone makes every hit non-black, a miss is white"
    st = rt.last_stats()
    assert st["rays_primary"] == 4 * (2 * (w // 2)) * (2 * (h // 2) - 1) and st["kernel_ms"] > 0
    return a


@pytest.fixture(scope="module")
def syn():
    return importlib.import_module("rust-ray-tracer_amd.synthetic")


@pytest.fixture(scope="module")
def soup100k(rrt, ob, syn):
    sd = rrt.parse_obj_file(syn.ensure_soup(ASSETS, 100000, syn.SEED_100K))
    assert sd.info["n_tris"] == 100000 and sd.info["n_nodes"] == 140265 and sd.info["root_own_count"] == 1089   # record the structure
    lights = rrt.default_lights()
    return sd, rrt.RayTracer(sd, lights), oracle_scene_for(ob, rrt, sd, lights), lights


@pytest.fixture(scope="module")
def soup1m(rrt, ob, syn):
    sd = rrt.parse_obj_file(syn.ensure_soup(ASSETS, 1000000, syn.SEED_1M))
    assert sd.info["n_tris"] == 1000000 and sd.info["n_nodes"] == 818353 and sd.info["root_own_count"] == 10961
    lights = rrt.default_lights()
    return sd, rrt.RayTracer(sd, lights), oracle_scene_for(ob, rrt, sd, lights), lights


def test_config2_soup100k_1080p(rrt, soup100k):
    """configs[2]: 100 k-triangle soup @1920x1080."""
    sd, rt, osc, lights = soup100k
    w, h = 1920, 1080
    a = frame_invariants(rt, w, h)
    assert ((a != 0xFFFFFF) & (a != 0)).mean() > 0.5                                        # the soup fills most of the view
    rng = np.random.default_rng(102)
    rows = sorted(set(rng.integers(1, h, 6).tolist()) | {1, h // 2, h - 1})
    check_rows_against_oracle(a, osc, w, h, rows, 16)
    O, D, M = sample_rays(osc, w, h, 2500, rng, lights)
    check_rays_bit_exact(rt, osc, O, D, M, 5000, 0.3)
    exact = rrt.RayTracer(sd, lights, no_cull=True)
    assert np.array_equal(exact.render(w, h), a), "default (indexed) frame differs from the reference-order (no_cull) frame"
    for mode in ("lane", "bundle", "ray"):
        assert np.array_equal(rrt.RayTracer(sd, lights, box_filter=mode).render(w, h), a), mode
    # the reference's progressive display (50-row bands, engine.rs:196-253): band launches under the XCD-aware block order (a large scene) give the same frame
    assert np.array_equal(rt.render_progressive(w, h, chunk_rows=50), a)


def test_config3_teapot_4k(rrt, teapot, teapot_oracle):
    """configs[3] on one GPU: model2.obj @3840x2160 (the 8-GPU form of it is the tile partition, test_tile_partition_4k below)."""
    lights = rrt.default_lights()
    rt = rrt.RayTracer(teapot, lights)
    w, h = 3840, 2160
    a = frame_invariants(rt, w, h)
    rng = np.random.default_rng(103)
    rows = sorted(set(rng.integers(1, h, 6).tolist()) | {1, h // 2, h - 1})
    check_rows_against_oracle(a, teapot_oracle, w, h, rows, 16)
    O, D, M = sample_rays(teapot_oracle, w, h, 3000, rng, lights)
    check_rays_bit_exact(rt, teapot_oracle, O, D, M, 5000, 0.3)
    assert np.array_equal(rrt.RayTracer(teapot, lights, no_cull=True).render(w, h), a)


def test_config3_tile_partition_4k(rrt, teapot):
    """configs[3]'s partition at its real size: the 8 ranks' tile buffers of the 3840x2160 frame, rendered in turn on this GPU, gathered and de-tiled,
    equal the single-launch frame bit for bit (the RCCL gather itself moves bytes; rrt_render_multi's test covers it with one rank)."""
    torch = pytest.importorskip("torch")
    rt = rrt.RayTracer(teapot, rrt.default_lights())
    w, h, world = 3840, 2160, 8
    full = rt.render(w, h)
    tpr = rrt.tiles_per_rank(w, h, world)
    gathered = torch.empty((world, tpr * 64), dtype=torch.int32, device="cuda")
    for r in range(world):
        rt.render_tiles_into(gathered[r], w, h, r, world)
    fb = torch.empty((h, w), dtype=torch.int32, device="cuda")
    rt.detile_into(gathered, fb, w, h, world)
    torch.cuda.synchronize()
    assert np.array_equal(fb.cpu().numpy().view(np.uint32), full)


def test_config4_tile_partition_with_the_xcd_aware_block_order(rrt, soup1m):
    """configs[4]'s partition: the 8 ranks' tile buffers of the 1 M-triangle soup -- a scene large enough for the XCD-aware block order (render.hip:
    render_kernel, FrameParams::xcd_chunk), which re-labels the blocks of a launch -- rendered in turn on this GPU at 1920x1080 (8 100 local tiles per rank:
    whole chunks plus a tail), gathered and de-tiled: the single-launch frame bit for bit; and a size whose launch is smaller than one round of chunks."""
    torch = pytest.importorskip("torch")
    sd, rt, osc, lights = soup1m
    for w, h, world in ((1920, 1080, 8), (200, 120, 3)):
        full = rt.render(w, h)
        tpr = rrt.tiles_per_rank(w, h, world)
        gathered = torch.empty((world, tpr * 64), dtype=torch.int32, device="cuda")
        for r in range(world):
            rt.render_tiles_into(gathered[r], w, h, r, world)
        fb = torch.empty((h, w), dtype=torch.int32, device="cuda")
        rt.detile_into(gathered, fb, w, h, world)
        torch.cuda.synchronize()
        assert np.array_equal(fb.cpu().numpy().view(np.uint32), full), (w, h, world)


def test_config4_soup1m_4k(rrt, soup1m):
    """configs[4] on one GPU: 1 M-triangle soup @3840x2160 (root list of 10 961 straddlers -> group records, clusters.cpp)."""
    sd, rt, osc, lights = soup1m
    w, h = 3840, 2160
    a = frame_invariants(rt, w, h)
    assert ((a != 0xFFFFFF) & (a != 0)).mean() > 0.5
    rng = np.random.default_rng(104)
    rows = sorted(set(rng.integers(1, h, 2).tolist()) | {1, h // 2, h - 1})
    check_rows_against_oracle(a, osc, w, h, rows, 48)
    O, D, M = sample_rays(osc, w, h, 2500, rng, lights)
    check_rays_bit_exact(rt, osc, O, D, M, 5000, 0.3)
    # (d) reference-order mode on a 64-row band: every sub-sample ray of the band through rrt_get_ray_colours (primary + its shadow and
    # reflection rays) in both modes, and the default mode's band equals the frame's rows after Color::mix
    exact = rrt.RayTracer(sd, lights, no_cull=True)
    r0 = h // 2 - 32
    xs = np.arange(0, w)
    d = np.concatenate([row_dirs(w, h, r, xs).reshape(-1, 3) for r in range(r0, r0 + 64)])
    o = np.tile(ORIGIN, (len(d), 1))
    c_fast = rt.get_ray_colours(o, d); c_exact = exact.get_ray_colours(o, d)
    assert np.array_equal(c_fast, c_exact), f"{(c_fast != c_exact).sum()} of {len(d)} band rays differ between the indexed and the reference-order mode"
    mixed = channels(c_fast.reshape(64, 4, w)).sum(1) // 4
    assert np.array_equal(mixed, channels(a[r0:r0 + 64])), "band rays mixed per pixel differ from the frame's rows"


def _plane_scene(rng, apex, n_planes, per, n_filler):
    """Triangles constructed in f64 INSIDE planes through `apex` (the generator of the `noise` scene of test_gpu_parity.py, scaled up) + random filler
    triangles + a backdrop.  Returns triangles, and per plane (d0, u): the in-plane directions are d0 + s*u."""
    tris, planes = [], []
    for k in range(n_planes):
        d0 = np.array([rng.uniform(-0.45, 0.45), rng.uniform(-0.3, 0.3), 1.0]); u = rng.normal(size=3)
        for j in range(per):
            a0, a1, a2 = rng.uniform(6, 14, 3); b0, b1, b2 = rng.uniform(-3, 3, 3)
            tris.append([apex + a0 * d0 + b0 * u, apex + a1 * d0 + b1 * u, apex + a2 * d0 + b2 * u])
        planes.append((d0, u))
    for k in range(n_filler):
        p = rng.uniform([-4, -0.5, -3], [4, 5, 7]); sz = 10 ** rng.uniform(-1.5, 0.0)
        tris.append([p, p + rng.normal(size=3) * sz, p + rng.normal(size=3) * sz])
    tris.append([(-8, -2, 16), (8, -2, 16), (0, 9, 16.5)])
    return np.asarray(tris, np.float64), planes


def _coplanar_rays(rng, tris, planes, n_plane_tris, apex, N):
    """N rays from `apex`: in-plane directions (coplanar with that plane's triangles up to rounding), rays through vertices and through edge points of
    the in-plane triangles, each with and without a tiny perturbation (1e-16 .. 1e-9)."""
    pk = rng.integers(0, len(planes), N)
    d0s = np.array([p[0] for p in planes]); us = np.array([p[1] for p in planes])
    D = d0s[pk] + rng.uniform(-0.35, 0.35, N)[:, None] * us[pk]
    third = N // 3
    ti = rng.integers(0, n_plane_tris, third); vi = rng.integers(0, 3, third)
    D[:third] = tris[ti, vi] - apex
    w2 = rng.random((third, 1))
    D[third:2 * third] = (tris[ti, vi] * w2 + tris[ti, (vi + 1) % 3] * (1 - w2)) - apex
    eps = 10.0 ** rng.uniform(-16, -9, N) * (rng.random(N) < 0.5)
    D += rng.normal(size=(N, 3)) * eps[:, None]
    return np.tile(apex, (N, 1)), D


def _scene_data(rrt, tris):
    n = len(tris)
    nrm = np.tile([0.0, 0.1, -1.0], (n, 3, 1)); uv = np.tile([[0.1, 0.2, 0], [0.9, 0.1, 0], [0.5, 0.8, 0]], (n, 1, 1)).astype(np.float64)
    mats = [dict(ka=(1, 1, 1), kd=(1, 1, 1), ks=(1, 1, 1), ns=240.0, kr=0.0, tex=0, bump=-1)]
    return rrt.SceneData.from_arrays(tris, uv, nrm, np.zeros(n, np.uint32), mats, [np.arange(48, dtype=np.uint8).reshape(4, 4, 3)])


def _in_noise_band(tris, o, d, pad):
    """Host restatement of the exactness criterion (clusters.cpp, find_origin_suspects) for ONE ray against every triangle: is the ray's origin within
    delta of a triangle's plane AND its direction within alpha of parallel to it?  Outside that band the index is provably exact."""
    e1 = tris[:, 1] - tris[:, 0]; e2 = tris[:, 2] - tris[:, 0]; s = o - tris[:, 0]
    n = np.cross(e1, e2); ln = np.linalg.norm(n, axis=1); l1 = np.linalg.norm(e1, axis=1); l2 = np.linalg.norm(e2, axis=1)
    R = np.linalg.norm(s, axis=1) + np.maximum(l1, l2); sinphi = ln / (l1 * l2); eps = 2.0 ** -53
    alpha = 8 * 64 * eps * R / (pad * sinphi); delta = 2 * (alpha * R + 64 * eps * R) / sinphi
    rho = np.abs((s * n).sum(1)) / ln; sina = np.abs(n @ d) / (ln * np.linalg.norm(d))
    return bool(((rho <= delta) & (sina <= alpha)).any())


def test_filter_exactness_guard_near_coplanar_rays_at_scale(rrt):
    """The fp32 box filters assume that a (ray, triangle) pair the reference's f64 Moller-Trumbore accepts lies within the padded boxes; that fails only
    for a ray lying IN a triangle's plane to rounding noise.  Rays from the raytracer's origin -- every primary ray -- are guarded:
    triangles whose plane contains the origin are found at create time and a ray within their noise band runs unfiltered.
      A. 60 such triangles (6 planes, under the cap of 64) among 3000 others, 10^6 rays from the origin, half of them coplanar-to-rounding with those
         triangles (the `noise` generator at scale), through their vertices and edges: lane filter, bundle filter and the autotuned default must return
         (hit, t, u, v, triangle) bit for bit as the reference-order mode does.  [Without the guard ~3 % of such rays differ.]
      B. 4000 such triangles (over the cap: every ray from the origin runs unfiltered), 2*10^5 rays: same.
      C. the same construction around ANOTHER apex (rays that do not start at the raytracer's origin: the stated caveat): every ray whose result differs
         from the reference-order mode must lie inside the noise band of some triangle (origin within delta of its plane AND direction within alpha)."""
    rng = np.random.default_rng(55)
    o0 = np.array(ORIGIN)
    lights = rrt.default_lights()
    # --- A
    tris, planes = _plane_scene(rng, o0, 6, 10, 3000)
    sd = _scene_data(rrt, tris)
    N = 1_000_000
    O, D = _coplanar_rays(rng, tris, planes, 60, o0, N)
    D[N // 2:] = np.stack([rng.uniform(-0.5, 0.5, N - N // 2), rng.uniform(-0.35, 0.35, N - N // 2), np.ones(N - N // 2)], -1)   # ordinary rays beside them
    exact = rrt.RayTracer(sd, lights, no_cull=True).intersect_rays(O, D)
    assert 0.2 < exact[0].mean() <= 1.0
    for mode in ("lane", "bundle", "ray", None):
        rt = rrt.RayTracer(sd, lights, box_filter=mode)
        assert rt.last_stats()["origin_plane_triangles"] == 60
        got = rt.intersect_rays(O, D)
        for name, x, y in zip(("hit", "t", "u", "v", "tri"), got, exact):
            bad = x != y
            assert not bad.any(), f"A, filter {mode}: {name} differs from the reference-order mode on {bad.sum()} of {N} rays (first: ray {int(np.argmax(bad))})"
        fa = rt.render(256, 192)
        assert np.array_equal(fa, rrt.RayTracer(sd, lights, no_cull=True).render(256, 192)), f"A, filter {mode}: frame differs"
    # --- B
    tris, planes = _plane_scene(rng, o0, 400, 10, 0)
    sd = _scene_data(rrt, tris)
    N = 200_000
    O, D = _coplanar_rays(rng, tris, planes, 4000, o0, N)
    exact = rrt.RayTracer(sd, lights, no_cull=True).intersect_rays(O, D)
    for mode in ("lane", "bundle", "ray"):
        rt = rrt.RayTracer(sd, lights, box_filter=mode)
        assert rt.last_stats()["origin_plane_triangles"] == 4000
        for name, x, y in zip(("hit", "t", "u", "v", "tri"), rt.intersect_rays(O, D), exact):
            assert np.array_equal(x, y), f"B, filter {mode}: {name} differs"
    # --- C: rays from another apex (not the raytracer's origin): differences only inside the noise band
    apex = np.array([1.5, 1.0, -8.0])
    tris, planes = _plane_scene(rng, apex, 40, 10, 1000)
    sd = _scene_data(rrt, tris)
    N = 200_000
    O, D = _coplanar_rays(rng, tris, planes, 400, apex, N)
    exact = rrt.RayTracer(sd, lights, no_cull=True).intersect_rays(O, D)
    rt = rrt.RayTracer(sd, lights, box_filter="lane")
    assert rt.last_stats()["origin_plane_triangles"] == 0
    got = rt.intersect_rays(O, D)
    differ = np.zeros(N, bool)
    for x, y in zip(got, exact):
        differ |= x != y
    pad = 20.0 / 32768.0                                           # clusters.cpp: 2^-15 of the scene magnitude (root box +-20)
    idx = np.flatnonzero(differ)
    for i in idx[:3000]:
        assert _in_noise_band(tris[:-1], O[i], D[i], pad), f"C: ray {i} differs from the reference-order mode although it is outside every triangle's noise band"
    print(f"caveat C: {len(idx)} of {N} constructed in-plane rays from a foreign apex differ (all inside the noise band)")
