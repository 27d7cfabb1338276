"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/rrt_oracle.c).  Run from the repo root:
    python tests/golden/make_golden.py
The reference cannot be built or run here (no Rust toolchain), so these vectors are ORACLE outputs: they pin the oracle
against regressions and give the GPU tests fixed inputs/outputs; they are not outputs of the reference binary."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob   # noqa: E402

rrt = importlib.import_module("rust-ray-tracer_amd")   # host-side loader only (no GPU needed)
OUT = os.path.dirname(os.path.abspath(__file__))


def scene(name):
    sd = rrt.parse_obj_file(os.path.join(ROOT, "assets", name))
    pos, uv, nrm, mat = sd.triangles()
    lights = [(l.kind, l.intensity, (l.v.x, l.v.y, l.v.z)) for l in rrt.default_lights()]
    return sd, ob.OracleScene(pos, uv, nrm, mat, sd.materials(), sd.textures(), lights, (0.0, 2.0, -10.0))


def main():
    for name, sizes in (("model2.obj", [(64, 48), (97, 61), (160, 120)]), ("model.obj", [(64, 48)]), ("model3.obj", [(64, 48)])):
        sd, osc = scene(name)
        frames = {f"fb_{w}x{h}": osc.render(w, h)[0] for (w, h) in sizes}
        # fixed rays: primary rays of a 32x24 grid + pseudo-random secondary-like rays starting inside the scene
        rng = np.random.default_rng(12345)
        n = 512
        o = np.concatenate([np.tile([0.0, 2.0, -10.0], (n, 1)), rng.uniform([-3, 0.2, -3], [3, 4, 3], (n, 3))])
        d = np.concatenate([np.stack([rng.uniform(-0.5, 0.5, n), rng.uniform(-0.5, 0.5, n), np.ones(n)], -1), rng.normal(size=(n, 3))])
        hit = np.zeros(2 * n, bool); t = np.zeros(2 * n); u = np.zeros(2 * n); v = np.zeros(2 * n); tri = np.zeros(2 * n, np.uint32)
        col = np.zeros(2 * n, np.uint32)
        for i in range(2 * n):
            hit[i], t[i], u[i], v[i], tri[i] = osc.intersect(o[i], d[i])
            col[i] = osc.get_ray_colour(o[i], d[i])
        tree = osc.octree()
        np.savez_compressed(os.path.join(OUT, name.replace(".obj", "") + ".npz"), ray_o=o, ray_d=d, ray_hit=hit, ray_t=t, ray_u=u, ray_v=v,
                            ray_tri=tri, ray_col=col, n_nodes=len(tree["first_child"]), root_own=int(tree["own_off"][1]),
                            max_depth=tree["max_depth"], tri_count_sum=int(tree["tri_count"].sum()), **frames)
        print(name, "done")


if __name__ == "__main__":
    main()
