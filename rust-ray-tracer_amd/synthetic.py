"""Deterministic synthetic triangle soups for the large-scene configs of BASELINE.json (the reference ships none).

Recipe: splitmix64 stream; per triangle 18 draws in this order -- centre (cx,cy,cz) uniform in
[-4.5,4.5] x [0.5,5.5] x [-4.5,4.5] (in view of the fixed camera, inside the +-20 root box), three vertex offsets
uniform in [-s,s]^3 (s = 0.05), three (u,v) texture coordinates uniform in [0,1)^2; vn = the face normal; material
`teapot`.  The scene is emitted as .obj TEXT with 6 decimals so that every consumer parses identical f64 values in the
same order (triangle order decides the octree, octree.rs:41-108).
"""
from __future__ import annotations

import os

import numpy as np

SEED_100K = 0x5EED0001
SEED_1M = 0x5EED0002
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(seed: int, n: int) -> np.ndarray:
    """First n outputs of splitmix64 seeded with `seed` (vectorised: state_i = seed + (i+1)*golden)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + _GOLDEN * np.arange(1, n + 1, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def soup_arrays(n_tris: int, seed: int, s: float = 0.05):
    u = (splitmix64(seed, 18 * n_tris) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    u = u.reshape(n_tris, 18)
    centre = np.stack([-4.5 + 9.0 * u[:, 0], 0.5 + 5.0 * u[:, 1], -4.5 + 9.0 * u[:, 2]], -1)
    verts = centre[:, None, :] + (2.0 * u[:, 3:12].reshape(n_tris, 3, 3) - 1.0) * s
    vt = u[:, 12:18].reshape(n_tris, 3, 2)
    nrm = np.cross(verts[:, 1] - verts[:, 0], verts[:, 2] - verts[:, 0])
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-300)
    return verts, vt, nrm


def _faces(lo: int, hi: int) -> np.ndarray:
    i = np.arange(lo, hi, dtype=np.int64)
    return np.stack([3 * i + 1, 3 * i + 1, i + 1, 3 * i + 2, 3 * i + 2, i + 1, 3 * i + 3, 3 * i + 3, i + 1], -1)


def _write_part(base: str, n_tris: int, seed: int, k: int, parts: int) -> None:
    """Worker k of `parts`: the text of triangles [lo, hi) of each of the four sections, one part file per section."""
    verts, vt, nrm = soup_arrays(n_tris, seed)
    lo, hi = n_tris * k // parts, n_tris * (k + 1) // parts
    np.savetxt(f"{base}.v.{k}", verts[lo:hi].reshape(-1, 3), fmt="v %.6f %.6f %.6f")
    np.savetxt(f"{base}.vt.{k}", vt[lo:hi].reshape(-1, 2), fmt="vt %.6f %.6f")
    np.savetxt(f"{base}.vn.{k}", nrm[lo:hi], fmt="vn %.6f %.6f %.6f")
    np.savetxt(f"{base}.f.{k}", _faces(lo, hi), fmt="f %d/%d/%d %d/%d/%d %d/%d/%d")


def write_soup_obj(path: str, n_tris: int, seed: int, mtllib: str = "materials.mtl", material: str = "teapot", workers: int | None = None) -> str:
    """Writes the soup as a .obj next to `mtllib` (texture names resolve against the .obj's directory).  Formatting 12 M numbers is the slow part,
    so large soups are formatted by `workers` fresh interpreter processes (this file run as a script: numpy only, nothing that touches a GPU),
    each writing its slice of every section; the parts are concatenated in order, so the bytes do not depend on the worker count."""
    import shutil
    import subprocess
    import sys
    if workers is None:
        workers = 1 if n_tris < 200000 else max(1, min(16, (os.cpu_count() or 1)))
    tmp = path + f".tmp{os.getpid()}"
    if workers == 1:
        _write_part(tmp, n_tris, seed, 0, 1)
    else:
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), tmp, str(n_tris), str(seed), str(k), str(workers)]) for k in range(workers)]
        rcs = [p.wait() for p in procs]
        if any(rcs):
            raise RuntimeError(f"soup writer workers failed: {rcs}")
    with open(tmp, "wb") as f:
        f.write(f"# synthetic soup: {n_tris} triangles, splitmix64 seed {seed:#x}\nmtllib {mtllib}\n".encode())
        for sec in ("v", "vt", "vn", "f"):
            if sec == "f":
                f.write(f"usemtl {material}\n".encode())
            for k in range(workers):
                part = f"{tmp}.{sec}.{k}"
                with open(part, "rb") as g:
                    shutil.copyfileobj(g, f, 1 << 24)
                os.remove(part)
    os.replace(tmp, path)
    return path


def ensure_soup(assets_dir: str, n_tris: int, seed: int) -> str:
    path = os.path.join(assets_dir, f"soup_{n_tris}_{seed:x}.obj")
    if not os.path.exists(path):
        write_soup_obj(path, n_tris, seed)
    return path


if __name__ == "__main__":          # worker entry of write_soup_obj
    import sys
    _write_part(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
