import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
ASSETS = os.path.join(ROOT, "assets")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rrt():
    """The product package (hyphenated directory name -> importlib)."""
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(ROOT, "rust-ray-tracer_amd", "librrt_hip.so")):
        g.build()
    return importlib.import_module("rust-ray-tracer_amd")


@pytest.fixture(scope="session")
def ob():
    """The oracle binding (checker)."""
    from oracle import binding
    binding.lib()
    return binding


def lights_tuple(lights):
    return [(l.kind, l.intensity, (l.v.x, l.v.y, l.v.z)) for l in lights]


def oracle_scene_for(ob, rrt, sd, lights=None, origin=(0.0, 2.0, -10.0)):
    pos, uv, nrm, mat = sd.triangles()
    lights = rrt.default_lights() if lights is None else lights
    return ob.OracleScene(pos, uv, nrm, mat, sd.materials(), sd.textures(), lights_tuple(lights), origin)


@pytest.fixture(scope="session")
def teapot(rrt):
    return rrt.parse_obj_file(os.path.join(ASSETS, "model2.obj"))


@pytest.fixture(scope="session")
def teapot_oracle(ob, rrt, teapot):
    return oracle_scene_for(ob, rrt, teapot)


def channels(fb):
    fb = np.asarray(fb)
    return np.stack([(fb >> 16) & 255, (fb >> 8) & 255, fb & 255], -1).astype(np.int64)


def max_channel_diff(a, b):
    return int(np.abs(channels(a) - channels(b)).max())
