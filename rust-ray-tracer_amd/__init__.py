"""MI355X-native per-pixel hot path of conor722/rust-ray-tracer -- Python host mirror over the C ABI (include/rrt.h).

The names follow the reference's host code so tests read like the reference would test itself:

    SceneData  <- parse_obj_file_lines()          src/file_management/utils.rs:139, src/scene/scenedata.rs:5-13
    Light.*                                       src/scene/entities.rs:5-9
    RayTracer(scene_data, lights, origin)         src/scene/raytracer.rs:22-26   (.get_ray_colour -> raytracer.rs:29)
    Scene(width, height).draw_scene(rt)           src/scene/engine.rs:177,186    (fills scene.canvas.buffer, engine.rs:127)

Everything that computes goes through librrt_hip.so (hand-written HIP kernels, gfx950).  There is no CPU or
PyTorch fallback: if the library is missing, or no GPU is visible when a RayTracer is created, this raises.
PyTorch is only plumbing (device buffers, streams, torch.distributed) in `render_into` / `render_tiles_into`.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Iterable, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RRT_LIB") or os.path.join(_HERE, "librrt_hip.so")   # RRT_LIB: developer override (e.g. the counters build)


class RrtError(RuntimeError):
    def __init__(self, status: int, what: str, detail: str):
        super().__init__(f"{what}: {detail or '?'} (status {status})")
        self.status = status
        self.detail = detail


FLAG_NO_CULL, FLAG_LANE_FILTER, FLAG_BUNDLE_FILTER, FLAG_RAY_WALK, FLAG_HOST_SETUP = 1, 2, 4, 8, 16   # RRT_FLAG_*, include/rrt.h
BUFFERS = ("nodes", "geom", "attr", "supers", "cboxes", "child_boxes", "tboxes", "suspects", "oct_box", "oct_first_child", "oct_tri_count", "oct_own_off",
           "oct_own_idx", "slot_tri", "slot_pos")   # RRT_BUF_*
VARIANT_NAMES = ("lane", "bundle", "ray")   # rrt_stats.filter_variant

# status codes, include/rrt.h
OK, ERR_INVALID_ARG, ERR_HIP, ERR_OOM, ERR_IO, ERR_PARSE, ERR_DEPTH, ERR_NO_DEVICE, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6, -7, -8


class Vec3(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]


class CLight(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("_pad", C.c_uint32), ("intensity", C.c_double), ("v", Vec3)]


class CMaterial(C.Structure):
    _fields_ = [("ka", Vec3), ("kd", Vec3), ("ks", Vec3), ("ns", C.c_double), ("kr", C.c_double), ("tex", C.c_int32), ("bump", C.c_int32)]


class CTexture(C.Structure):
    _fields_ = [("rgb", C.POINTER(C.c_uint8)), ("width", C.c_uint32), ("height", C.c_uint32)]


class COptions(C.Structure):
    _fields_ = [("surface_offset", C.c_double), ("max_reflection_depth", C.c_uint32), ("flags", C.c_uint32),
                ("vp_w", C.c_double), ("vp_h", C.c_double), ("vp_d", C.c_double)]


class CModelInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("n_tris", "n_tris_in_tree", "n_nodes", "max_depth", "n_mats", "n_tex", "root_own_count", "max_own_count")]


class CStats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("width", C.c_uint32), ("height", C.c_uint32), ("rays_primary", C.c_uint64), ("scene_bytes", C.c_uint64),
                ("filter_variant", C.c_uint32), ("origin_plane_triangles", C.c_uint32),
                ("filter_pad", C.c_double), ("filter_alpha_unit", C.c_double), ("filter_delta_unit", C.c_double)]


class CSetupTimes(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("read_ms", "parse_ms", "texture_ms", "octree_ms", "index_ms", "upload_ms", "hip_init_ms", "create_ms", "gpu_setup")]


# every symbol include/rrt.h declares: (restype, argtypes)
_P = C.c_void_p
_dp, _u32p, _u8p = C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
SYMBOLS = {
    "rrt_model_load_obj": (C.c_int, [C.c_char_p, _dp, C.POINTER(_P)]),
    "rrt_model_from_arrays": (C.c_int, [C.c_uint32, _dp, _dp, _dp, _u32p, C.c_uint32, C.POINTER(CMaterial), C.c_uint32, C.POINTER(CTexture), _dp, C.POINTER(_P)]),
    "rrt_model_destroy": (None, [_P]),
    "rrt_model_get_info": (C.c_int, [_P, C.POINTER(CModelInfo)]),
    "rrt_model_get_triangles": (C.c_int, [_P, _dp, _dp, _dp, _u32p]),
    "rrt_model_get_materials": (C.c_int, [_P, C.POINTER(CMaterial)]),
    "rrt_model_get_texture": (C.c_int, [_P, C.c_uint32, C.POINTER(CTexture)]),
    "rrt_model_get_octree": (C.c_int, [_P, _dp, _u32p, _u32p, _u32p, _u32p]),
    "rrt_decode_image_file": (C.c_int, [C.c_char_p, C.POINTER(_u8p), _u32p, _u32p]),
    "rrt_free": (None, [_P]),
    "rrt_raytracer_create": (C.c_int, [_P, C.POINTER(CLight), C.c_uint32, Vec3, C.POINTER(COptions), C.c_int, C.POINTER(_P)]),
    "rrt_raytracer_create_from_arrays": (C.c_int, [C.c_uint32, _dp, _dp, _dp, _u32p, C.c_uint32, C.POINTER(CMaterial), C.c_uint32, C.POINTER(CTexture), _dp,
                                                  C.POINTER(CLight), C.c_uint32, Vec3, C.POINTER(COptions), C.c_int, C.POINTER(_P)]),
    "rrt_raytracer_destroy": (None, [_P]),
    "rrt_render": (C.c_int, [_P, C.c_uint32, C.c_uint32, _u32p]),
    "rrt_host_buffer_register": (C.c_int, [_P, C.c_size_t]),
    "rrt_host_buffer_unregister": (C.c_int, [_P]),
    "rrt_render_device": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P, _P]),
    "rrt_tiles_per_rank": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32]),
    "rrt_render_tiles_device": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _P, _P]),
    "rrt_detile_device": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32, _P, _P, _P]),
    "rrt_multi_create": (C.c_int, [C.POINTER(_P), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_P)]),
    "rrt_dist_unique_id": (C.c_int, [_P]),
    "rrt_dist_create": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P, C.c_uint32, C.POINTER(_P)]),
    "rrt_multi_destroy": (None, [_P]),
    "rrt_multi_enqueue": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P]),
    "rrt_multi_sync": (C.c_int, [_P]),
    "rrt_render_multi": (C.c_int, [_P, C.c_uint32, C.c_uint32, _u32p]),
    "rrt_multi_last_gather_ms": (C.c_int, [_P, _dp]),
    "rrt_render_progressive": (C.c_int, [_P, C.c_uint32, C.c_uint32, _u32p, C.c_uint32, _P, _P]),
    "rrt_get_ray_colours": (C.c_int, [_P, C.c_uint32, _dp, _dp, _u32p]),
    "rrt_intersect_rays": (C.c_int, [_P, C.c_uint32, _dp, _dp, _dp, _u8p, _dp, _dp, _dp, _u32p]),
    "rrt_raytracer_get_octree": (C.c_int, [_P, C.POINTER(CModelInfo), _dp, _u32p, _u32p, _u32p, _u32p]),
    "rrt_raytracer_get_buffer": (C.c_int, [_P, C.c_uint32, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "rrt_last_stats": (C.c_int, [_P, C.POINTER(CStats)]),
    "rrt_get_setup_times": (C.c_int, [_P, _P, C.POINTER(CSetupTimes)]),
    "rrt_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "rrt_strerror": (C.c_char_p, [C.c_int]),
    "rrt_last_error_detail": (C.c_char_p, []),
    "rrt_build_info": (C.c_char_p, []),
}

_lib = None


def lib() -> C.CDLL:
    """Load librrt_hip.so (built by __graft_entry__.build() / csrc/Makefile).  Fails loudly when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: the HIP extension was not built (run `python -c 'import __graft_entry__ as g; g.build()'`). "
                              "There is no CPU fallback.")
        # One HIP runtime per process.  The PyTorch wheel bundles its own libamdhip64.so (soname libamdhip64.so.7, the name this library needs), so
        # whichever is loaded FIRST serves both; loaded second, torch would bring up a second runtime by file path and find "No HIP GPUs".  Tests and
        # bench.py use torch for device buffers, so it goes first here (RRT_NO_TORCH_PRELOAD=1: a host without torch in the process).
        if os.environ.get("RRT_NO_TORCH_PRELOAD") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _check(status: int, what: str):
    if status != OK:
        L = lib()
        detail = (L.rrt_last_error_detail() or b"").decode(errors="replace")
        raise RrtError(status, f"{what} failed [{L.rrt_strerror(status).decode()}]", detail)


def _d(a: np.ndarray):
    return a.ctypes.data_as(_dp)


def device_count() -> int:
    n = C.c_int(0)
    _check(lib().rrt_device_count(C.byref(n)), "rrt_device_count")
    return n.value


# ---------------------------------------------------------------------------------------------- reference-shaped host types
@dataclass
class Vector3d:                      # src/scene/engine.rs:9-14
    x: float
    y: float
    z: float

    def _c(self) -> Vec3:
        return Vec3(float(self.x), float(self.y), float(self.z))


@dataclass
class Light:                         # src/scene/entities.rs:5-9
    kind: int
    intensity: float
    v: Vector3d

    @staticmethod
    def Ambient(intensity: float) -> "Light":
        return Light(0, intensity, Vector3d(0.0, 0.0, 0.0))

    @staticmethod
    def Point(intensity: float, position: Vector3d) -> "Light":
        return Light(1, intensity, position)

    @staticmethod
    def Directional(intensity: float, direction: Vector3d) -> "Light":
        return Light(2, intensity, direction)


def default_lights() -> list:
    """The lights `main` hard-codes, in its order (src/main.rs:32-58)."""
    return [Light.Ambient(0.5), Light.Point(0.4, Vector3d(-7.0, 1.0, -15.0)), Light.Point(0.5, Vector3d(0.0, 1.0, -41.0)),
            Light.Directional(0.4, Vector3d(-5.0, 0.0, 20.0))]


DEFAULT_ORIGIN = Vector3d(0.0, 2.0, -10.0)   # src/main.rs:62-66
DEFAULT_ROOT = (-20.0, 20.0, -20.0, 20.0, -20.0, 20.0)   # src/file_management/utils.rs:145


class _Arrays:
    pass


def _marshal_arrays(pos, uv, nrm, mat, materials, textures, root):
    """ctypes views of a scene held in numpy arrays (no copies of arrays that are already contiguous float64 / uint32 / uint8)."""
    a = _Arrays()
    a.pos = np.ascontiguousarray(pos, np.float64).reshape(-1, 9)
    a.uv = np.ascontiguousarray(uv, np.float64).reshape(-1, 9)
    a.nrm = np.ascontiguousarray(nrm, np.float64).reshape(-1, 9)
    a.mat = np.ascontiguousarray(mat, np.uint32)
    a.n = a.pos.shape[0]
    a.cm = (CMaterial * max(1, len(materials)))()
    for i, m in enumerate(materials):
        a.cm[i] = CMaterial(Vec3(*m["ka"]), Vec3(*m["kd"]), Vec3(*m["ks"]), float(m["ns"]), float(m["kr"]), int(m["tex"]), int(m.get("bump", -1)))
    a.keep = [np.ascontiguousarray(t, np.uint8) for t in textures]
    a.ct = (CTexture * max(1, len(a.keep)))()
    for i, t in enumerate(a.keep):
        a.ct[i] = CTexture(t.ctypes.data_as(_u8p), t.shape[1], t.shape[0])
    a.r = (C.c_double * 6)(*root)
    return a


class SceneData:
    """SceneData (scenedata.rs:5-13): triangles in push order + materials + decoded textures + the octree."""

    def __init__(self, handle: int):
        self._h = _P(handle)
        self._info = None

    @property
    def info(self) -> dict:
        """rrt_model_get_info.  Builds the HOST copy of the octree on first use (the default GPU set-up of a RayTracer never needs it), so this is
        also where a too-deep tree (RRT_ERR_DEPTH) is reported on the host side."""
        if self._info is None:
            info = CModelInfo()
            _check(lib().rrt_model_get_info(self._h, C.byref(info)), "rrt_model_get_info")
            self._info = {n: getattr(info, n) for n, _ in CModelInfo._fields_}
        return self._info

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.rrt_model_destroy(h)

    @staticmethod
    def from_arrays(pos, uv, nrm, mat, materials: Sequence[dict], textures: Sequence[np.ndarray], root=DEFAULT_ROOT) -> "SceneData":
        """pos/uv/nrm: [n,3,3] float64; mat: [n] uint32; materials: dicts ka,kd,ks,ns,kr,tex,bump; textures: [h,w,3] uint8."""
        a = _marshal_arrays(pos, uv, nrm, mat, materials, textures, root)
        out = _P()
        _check(lib().rrt_model_from_arrays(a.n, _d(a.pos), _d(a.uv), _d(a.nrm), a.mat.ctypes.data_as(_u32p), len(materials), a.cm, len(a.keep), a.ct, a.r, C.byref(out)),
               "rrt_model_from_arrays")
        return SceneData(out.value)

    # --- accessors
    def triangles(self):
        n = self.info["n_tris"]
        pos, uv, nrm, mat = (np.empty((n, 3, 3)), np.empty((n, 3, 3)), np.empty((n, 3, 3)), np.empty(n, np.uint32))
        _check(lib().rrt_model_get_triangles(self._h, _d(pos), _d(uv), _d(nrm), mat.ctypes.data_as(_u32p)), "rrt_model_get_triangles")
        return pos, uv, nrm, mat

    def materials(self) -> list:
        n = self.info["n_mats"]
        cm = (CMaterial * max(1, n))()
        _check(lib().rrt_model_get_materials(self._h, cm), "rrt_model_get_materials")
        v = lambda a: (a.x, a.y, a.z)
        return [dict(ka=v(m.ka), kd=v(m.kd), ks=v(m.ks), ns=m.ns, kr=m.kr, tex=m.tex, bump=m.bump) for m in cm[:n]]

    def texture(self, i: int) -> np.ndarray:
        t = CTexture()
        _check(lib().rrt_model_get_texture(self._h, i, C.byref(t)), "rrt_model_get_texture")
        return np.ctypeslib.as_array(t.rgb, shape=(t.height, t.width, 3)).copy()

    def textures(self) -> list:
        return [self.texture(i) for i in range(self.info["n_tex"])]

    def octree(self) -> dict:
        n = self.info["n_nodes"]
        aabb = np.empty((n, 6)); fc = np.empty(n, np.uint32); tc = np.empty(n, np.uint32); off = np.empty(n + 1, np.uint32)
        idx = np.empty(self.info["n_tris_in_tree"], np.uint32)
        u = lambda a: a.ctypes.data_as(_u32p)
        _check(lib().rrt_model_get_octree(self._h, _d(aabb), u(fc), u(tc), u(off), u(idx)), "rrt_model_get_octree")
        return dict(aabb=aabb, first_child=fc, tri_count=tc, own_off=off, own_idx=idx, max_depth=self.info["max_depth"])


def parse_obj_file(path: str, root=DEFAULT_ROOT) -> SceneData:
    """fs::read_to_string + parse_obj_file_lines (src/main.rs:28-30, src/file_management/utils.rs:139-213)."""
    r = (C.c_double * 6)(*root)
    out = _P()
    _check(lib().rrt_model_load_obj(os.fsencode(path), r, C.byref(out)), f"parse_obj_file({path})")
    return SceneData(out.value)


def decode_image_file(path: str) -> np.ndarray:
    """The build-owned stand-in for `image::ImageReader::open(..).decode()` (utils.rs:345-350): [h,w,3] uint8."""
    p = _u8p(); w = C.c_uint32(); h = C.c_uint32()
    _check(lib().rrt_decode_image_file(os.fsencode(path), C.byref(p), C.byref(w), C.byref(h)), f"decode_image_file({path})")
    try:
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    finally:
        lib().rrt_free(p)


class RayTracer:
    """RayTracer{scene_data, lights, origin} (raytracer.rs:22-26), uploaded once to one MI355X."""

    def __init__(self, scene_data: SceneData, lights: Iterable[Light], origin: Vector3d = DEFAULT_ORIGIN, device: int = 0,
                 surface_offset: float = 0.0001, max_reflection_depth: int = 5, viewport=(1.0, 1.0, 1.0), no_cull: bool = False,
                 box_filter: Optional[str] = None, host_setup: bool = False):
        """no_cull=True (RRT_FLAG_NO_CULL): walk every own list in full, in list order, as ray.rs:119-129; default uses the cluster boxes.
        box_filter: None = rule of thumb on the first frame of a size, measured on the second, "lane" / "bundle" / "ray" = forced (RRT_FLAG_LANE_FILTER / RRT_FLAG_BUNDLE_FILTER /
        RRT_FLAG_RAY_WALK); same pixels.  host_setup=True (RRT_FLAG_HOST_SETUP): octree, index and records built on the host and uploaded (default: built on
        the GPU, csrc/scene_build.hip); same bytes in HBM."""
        self.scene_data, self.lights, self.origin, self.device = scene_data, list(lights), origin, device
        cl = (CLight * max(1, len(self.lights)))()
        for i, l in enumerate(self.lights):
            cl[i] = CLight(l.kind, 0, float(l.intensity), l.v._c())
        flags = (FLAG_NO_CULL if no_cull else 0) | (FLAG_HOST_SETUP if host_setup else 0) | {None: 0, "lane": FLAG_LANE_FILTER, "bundle": FLAG_BUNDLE_FILTER, "ray": FLAG_RAY_WALK}[box_filter]
        opt = COptions(surface_offset, max_reflection_depth, flags, *map(float, viewport))
        out = _P()
        _check(lib().rrt_raytracer_create(scene_data._h, cl, len(self.lights), origin._c(), C.byref(opt), device, C.byref(out)), "rrt_raytracer_create")
        self._h = out

    @classmethod
    def from_arrays(cls, pos, uv, nrm, mat, materials: Sequence[dict], textures: Sequence[np.ndarray], lights: Iterable[Light], origin: Vector3d = DEFAULT_ORIGIN,
                    device: int = 0, root=DEFAULT_ROOT, no_cull: bool = False, box_filter: Optional[str] = None) -> "RayTracer":
        """rrt_raytracer_create_from_arrays: the raytracer straight from the host's arrays (no SceneData / rrt_model, no host copy of the scene)."""
        self = cls.__new__(cls)
        self.scene_data, self.lights, self.origin, self.device = None, list(lights), origin, device
        a = _marshal_arrays(pos, uv, nrm, mat, materials, textures, root)
        cl = (CLight * max(1, len(self.lights)))()
        for i, l in enumerate(self.lights):
            cl[i] = CLight(l.kind, 0, float(l.intensity), l.v._c())
        flags = (FLAG_NO_CULL if no_cull else 0) | {None: 0, "lane": FLAG_LANE_FILTER, "bundle": FLAG_BUNDLE_FILTER, "ray": FLAG_RAY_WALK}[box_filter]
        opt = COptions(0.0001, 5, flags, 1.0, 1.0, 1.0)
        out = _P()
        _check(lib().rrt_raytracer_create_from_arrays(a.n, _d(a.pos), _d(a.uv), _d(a.nrm), a.mat.ctypes.data_as(_u32p), len(materials), a.cm, len(a.keep), a.ct, a.r,
                                                      cl, len(self.lights), origin._c(), C.byref(opt), device, C.byref(out)), "rrt_raytracer_create_from_arrays")
        self._h = out
        return self

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.rrt_raytracer_destroy(h)

    @property
    def info(self) -> dict:
        """rrt_model_info of the octree this raytracer's GPU set-up built (no host-side tree is built for it)."""
        info = CModelInfo()
        _check(lib().rrt_raytracer_get_octree(self._h, C.byref(info), None, None, None, None, None), "rrt_raytracer_get_octree")
        return {k: getattr(info, k) for k, _ in CModelInfo._fields_}

    def octree(self) -> dict:
        """The octree this raytracer's GPU set-up built (rrt_raytracer_get_octree): same dict as SceneData.octree(), plus "info"."""
        info = CModelInfo()
        _check(lib().rrt_raytracer_get_octree(self._h, C.byref(info), None, None, None, None, None), "rrt_raytracer_get_octree")
        n = info.n_nodes
        aabb = np.empty((n, 6)); fc = np.empty(n, np.uint32); tc = np.empty(n, np.uint32); off = np.empty(n + 1, np.uint32)
        idx = np.empty(info.n_tris_in_tree, np.uint32)
        u = lambda a: a.ctypes.data_as(_u32p)
        _check(lib().rrt_raytracer_get_octree(self._h, None, _d(aabb), u(fc), u(tc), u(off), u(idx)), "rrt_raytracer_get_octree")
        return dict(aabb=aabb, first_child=fc, tri_count=tc, own_off=off, own_idx=idx, max_depth=info.max_depth,
                    info={k: getattr(info, k) for k, _ in CModelInfo._fields_})

    def buffer(self, name: str) -> np.ndarray:
        """Raw bytes of one scene buffer in HBM (rrt_raytracer_get_buffer; tests compare the GPU set-up with the host set-up)."""
        which = BUFFERS.index(name)
        nb = C.c_size_t(0)
        _check(lib().rrt_raytracer_get_buffer(self._h, which, None, 0, C.byref(nb)), "rrt_raytracer_get_buffer")
        out = np.empty(nb.value, np.uint8)
        _check(lib().rrt_raytracer_get_buffer(self._h, which, out.ctypes.data_as(_P), nb.value, None), "rrt_raytracer_get_buffer")
        return out

    # raytracer.rs:29, batched
    def get_ray_colours(self, origins, dirs) -> np.ndarray:
        o = np.ascontiguousarray(origins, np.float64).reshape(-1, 3); d = np.ascontiguousarray(dirs, np.float64).reshape(-1, 3)
        assert o.shape == d.shape
        out = np.empty(o.shape[0], np.uint32)
        _check(lib().rrt_get_ray_colours(self._h, o.shape[0], _d(o), _d(d), out.ctypes.data_as(_u32p)), "rrt_get_ray_colours")
        return out

    def get_ray_colour(self, origin: Vector3d, direction: Vector3d) -> int:
        return int(self.get_ray_colours([[origin.x, origin.y, origin.z]], [[direction.x, direction.y, direction.z]])[0])

    # ray.rs:96-168, batched
    def intersect_rays(self, origins, dirs, max_t=None):
        o = np.ascontiguousarray(origins, np.float64).reshape(-1, 3); d = np.ascontiguousarray(dirs, np.float64).reshape(-1, 3)
        n = o.shape[0]
        mt = None if max_t is None else np.ascontiguousarray(np.broadcast_to(np.asarray(max_t, np.float64), (n,)))
        hit = np.empty(n, np.uint8); t = np.empty(n); u = np.empty(n); v = np.empty(n); tri = np.empty(n, np.uint32)
        _check(lib().rrt_intersect_rays(self._h, n, _d(o), _d(d), None if mt is None else _d(mt), hit.ctypes.data_as(_u8p), _d(t), _d(u), _d(v),
                                        tri.ctypes.data_as(_u32p)), "rrt_intersect_rays")
        return hit.astype(bool), t, u, v, tri

    # engine.rs:196-253: chunked draw with an update after every chunk (on_update(fb, first_row, n_rows) stands in for canvas.update())
    def render_progressive(self, width: int, height: int, on_update=None, chunk_rows: int = 50) -> np.ndarray:
        fb = np.empty((height, width), np.uint32)
        UPD = C.CFUNCTYPE(None, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32)
        cb = UPD(lambda user, p, w, h, r0, n: on_update(fb, int(r0), int(n))) if on_update is not None else None
        _check(lib().rrt_render_progressive(self._h, width, height, fb.ctypes.data_as(_u32p), chunk_rows, C.cast(cb, _P) if cb is not None else None, None),
               "rrt_render_progressive")
        return fb

    # engine.rs:186 via the C ABI, host framebuffer
    def render(self, width: int, height: int) -> np.ndarray:
        fb = np.empty((height, width), np.uint32)
        _check(lib().rrt_render(self._h, width, height, fb.ctypes.data_as(_u32p)), "rrt_render")
        return fb

    # device-resident variants (torch tensors are plumbing: data_ptr + current stream)
    def render_into(self, fb_tensor, width: int, height: int, stream: Optional[int] = None):
        assert fb_tensor.is_cuda and fb_tensor.is_contiguous() and fb_tensor.numel() == width * height and fb_tensor.element_size() == 4
        _check(lib().rrt_render_device(self._h, width, height, _P(fb_tensor.data_ptr()), _P(_stream(stream))), "rrt_render_device")

    def render_tiles_into(self, tiles_tensor, width: int, height: int, rank: int, world: int, stream: Optional[int] = None):
        need = tiles_per_rank(width, height, world) * 64
        assert tiles_tensor.is_cuda and tiles_tensor.is_contiguous() and tiles_tensor.numel() == need and tiles_tensor.element_size() == 4
        _check(lib().rrt_render_tiles_device(self._h, width, height, rank, world, _P(tiles_tensor.data_ptr()), _P(_stream(stream))), "rrt_render_tiles_device")

    def detile_into(self, gathered_tensor, fb_tensor, width: int, height: int, world: int, stream: Optional[int] = None):
        assert gathered_tensor.numel() == tiles_per_rank(width, height, world) * 64 * world and fb_tensor.numel() == width * height
        _check(lib().rrt_detile_device(self._h, width, height, world, _P(gathered_tensor.data_ptr()), _P(fb_tensor.data_ptr()), _P(_stream(stream))),
               "rrt_detile_device")

    # pre-bound launchers for per-frame loops (bench.py): all argument conversion is done once, the returned callable is one ctypes call
    def bind_render(self, fb_tensor, width: int, height: int, stream: Optional[int] = None):
        assert fb_tensor.is_cuda and fb_tensor.is_contiguous() and fb_tensor.numel() == width * height and fb_tensor.element_size() == 4
        fn, h, w_, h_, p, st = lib().rrt_render_device, self._h, C.c_uint32(width), C.c_uint32(height), _P(fb_tensor.data_ptr()), _P(_stream(stream))

        def launch():
            rc = fn(h, w_, h_, p, st)
            if rc != OK:
                _check(rc, "rrt_render_device")
        return launch

    def bind_render_tiles(self, tiles_tensor, width: int, height: int, rank: int, world: int, stream: Optional[int] = None):
        assert tiles_tensor.is_cuda and tiles_tensor.is_contiguous() and tiles_tensor.numel() == tiles_per_rank(width, height, world) * 64
        fn, h, st = lib().rrt_render_tiles_device, self._h, _P(_stream(stream))
        args = (C.c_uint32(width), C.c_uint32(height), C.c_uint32(rank), C.c_uint32(world), _P(tiles_tensor.data_ptr()))

        def launch():
            rc = fn(h, *args, st)
            if rc != OK:
                _check(rc, "rrt_render_tiles_device")
        return launch

    def bind_detile(self, gathered_tensor, fb_tensor, width: int, height: int, world: int, stream: Optional[int] = None):
        assert gathered_tensor.numel() == tiles_per_rank(width, height, world) * 64 * world and fb_tensor.numel() == width * height
        fn, h, st = lib().rrt_detile_device, self._h, _P(_stream(stream))
        args = (C.c_uint32(width), C.c_uint32(height), C.c_uint32(world), _P(gathered_tensor.data_ptr()), _P(fb_tensor.data_ptr()))

        def launch():
            rc = fn(h, *args, st)
            if rc != OK:
                _check(rc, "rrt_detile_device")
        return launch

    def setup_times(self) -> dict:
        """Wall ms of the once-per-scene stages: read, parse, texture decode, octree (model) + index, upload (this raytracer)."""
        t = CSetupTimes()
        _check(lib().rrt_get_setup_times(self.scene_data._h if self.scene_data is not None else None, self._h, C.byref(t)), "rrt_get_setup_times")
        return {n: getattr(t, n) for n, _ in CSetupTimes._fields_}

    def render_registered(self, width: int, height: int, fb: Optional[np.ndarray] = None) -> np.ndarray:
        """rrt_render into a page-locked framebuffer (rrt_host_buffer_register): the frame arrives by one asynchronous DMA."""
        if fb is None:
            fb = np.empty((height, width), np.uint32)
        p = _P(fb.ctypes.data)
        _check(lib().rrt_host_buffer_register(p, fb.nbytes), "rrt_host_buffer_register")
        try:
            _check(lib().rrt_render(self._h, width, height, fb.ctypes.data_as(_u32p)), "rrt_render")
        finally:
            _check(lib().rrt_host_buffer_unregister(p), "rrt_host_buffer_unregister")
        return fb

    def last_stats(self) -> dict:
        s = CStats()
        _check(lib().rrt_last_stats(self._h, C.byref(s)), "rrt_last_stats")
        return {n: getattr(s, n) for n, _ in CStats._fields_}


MULTI_LOOPBACK = 1   # RRT_MULTI_LOOPBACK


class MultiGpu:
    """The N GPUs of one node behind one handle (include/rrt.h, rrt_multi): the screen-tile partition, the RCCL gather to rank 0 and the de-tiling
    all happen inside the library.  MultiGpu(raytracers) = one process driving every GPU (rrt_multi_create); MultiGpu.dist(rt, rank, world, unique_id)
    = one process per GPU (rrt_dist_create; rank 0 makes the id with MultiGpu.unique_id() and the caller broadcasts it)."""

    def __init__(self, raytracers: Sequence["RayTracer"], frames_in_flight: int = 1, loopback: bool = False, _handle=None):
        self._keep = list(raytracers)
        if _handle is not None:
            self._h = _handle
            return
        arr = (_P * len(self._keep))(*[rt._h for rt in self._keep])
        out = _P()
        _check(lib().rrt_multi_create(arr, len(self._keep), frames_in_flight, MULTI_LOOPBACK if loopback else 0, C.byref(out)), "rrt_multi_create")
        self._h = out

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _check(lib().rrt_dist_unique_id(buf), "rrt_dist_unique_id")
        return buf.raw

    @staticmethod
    def dist(rt: "RayTracer", rank: int, world: int, unique_id: Optional[bytes], frames_in_flight: int = 1) -> "MultiGpu":
        out = _P()
        idbuf = C.create_string_buffer(unique_id, 128) if unique_id is not None else None
        _check(lib().rrt_dist_create(rt._h, rank, world, idbuf, frames_in_flight, C.byref(out)), "rrt_dist_create")
        return MultiGpu([rt], _handle=out)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.rrt_multi_destroy(h)

    def render(self, width: int, height: int) -> np.ndarray:
        """Blocking Scene::draw_scene over all GPUs, host framebuffer (rrt_render_multi)."""
        fb = np.empty((height, width), np.uint32)
        _check(lib().rrt_render_multi(self._h, width, height, fb.ctypes.data_as(_u32p)), "rrt_render_multi")
        return fb

    def bind_enqueue(self, fb_tensor, width: int, height: int):
        """One ctypes call per frame: trace -> gather -> de-tile enqueued on the next slot (fb_tensor on rank 0's GPU, None elsewhere)."""
        fn, h, w_, h_ = lib().rrt_multi_enqueue, self._h, C.c_uint32(width), C.c_uint32(height)
        p = _P(fb_tensor.data_ptr()) if fb_tensor is not None else _P()

        def enqueue():
            rc = fn(h, w_, h_, p)
            if rc != OK:
                _check(rc, "rrt_multi_enqueue")
        return enqueue

    def sync(self) -> None:
        _check(lib().rrt_multi_sync(self._h), "rrt_multi_sync")

    def last_gather_ms(self) -> float:
        v = C.c_double(-1.0)
        _check(lib().rrt_multi_last_gather_ms(self._h, C.byref(v)), "rrt_multi_last_gather_ms")
        return v.value


def _stream(stream: Optional[int]) -> int:
    if stream is not None:
        return stream
    import torch
    return torch.cuda.current_stream().cuda_stream


def tiles_per_rank(width: int, height: int, world: int) -> int:
    return int(lib().rrt_tiles_per_rank(width, height, world))


def tile_owner_map(width: int, height: int, world: int) -> np.ndarray:
    """[tiles_y, tiles_x] rank owning each 8x8-pixel tile (tile k -> k % world) -- host mirror of the kernel's partition."""
    tx, ty = (width + 7) // 8, (height + 7) // 8
    return (np.arange(tx * ty, dtype=np.int64) % world).reshape(ty, tx)


def detile_host(gathered: np.ndarray, width: int, height: int, world: int) -> np.ndarray:
    """Host mirror of rrt_detile_device (used by the gloo tests): gathered[world, tiles_per_rank, 64] -> fb[height, width]."""
    tx, ty = (width + 7) // 8, (height + 7) // 8
    tpr = (tx * ty + world - 1) // world
    g = np.asarray(gathered, np.uint32).reshape(world, tpr, 8, 8)
    k = np.arange(tx * ty)
    tiles = g[k % world, k // world].reshape(ty, tx, 8, 8)
    return tiles.transpose(0, 2, 1, 3).reshape(ty * 8, tx * 8)[:height, :width].copy()


class Canvas:                        # src/scene/engine.rs:123-167 minus the minifb window
    def __init__(self, width: int, height: int, on_update=None):
        self.width, self.height = width, height
        self.buffer = np.zeros((height, width), np.uint32)   # engine.rs:135
        self.on_update = on_update                           # stands in for window.update_with_buffer (engine.rs:162-166); None = no display
        self.updates = 0

    def update(self) -> None:                                # engine.rs:160-167
        self.updates += 1
        if self.on_update is not None:
            self.on_update(self.buffer)


class Scene:                         # src/scene/engine.rs:171-256
    def __init__(self, width: int, height: int, on_update=None):
        self.canvas = Canvas(width, height, on_update)

    def draw_scene(self, rt: RayTracer, progressive: bool = False) -> None:
        """Scene::draw_scene (engine.rs:186).  Default: one HIP launch instead of the rayon row loop, one canvas.update() at the end.
        progressive=True keeps the reference's pacing (engine.rs:196-253): 50 scene rows per chunk, canvas.update() after each."""
        if not progressive:
            self.canvas.buffer = rt.render(self.canvas.width, self.canvas.height)
            self.canvas.update()
            return

        def on_chunk(fb, first_row, n_rows):
            self.canvas.buffer = fb
            self.canvas.update()
        self.canvas.buffer = rt.render_progressive(self.canvas.width, self.canvas.height, on_chunk)
