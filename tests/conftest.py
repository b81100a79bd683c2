import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def rt():
    return importlib.import_module("rust-raytracer_amd")


@pytest.fixture(scope="session", autouse=True)
def _models():
    # procedural meshes are generated, not committed
    gen = importlib.import_module("scenes.gen_scenes")
    gen.ensure_models()


@pytest.fixture(scope="session")
def scene_path():
    def _p(name):
        return os.path.join(ROOT, "scenes", name + ".json")

    return _p


@pytest.fixture(scope="session")
def backend(rt):
    be = rt.Backend(0)
    yield be
    be.close()
