import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

SCENES = os.path.join(ROOT, "software-raytracer_amd", "scenes")
GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENE_NAMES = ["Scene1", "Scene1_reflection", "Scene2", "Scene3", "Scene3_indirect", "Scene_indirect"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import srt_oracle_py as O

    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def srt():
    return importlib.import_module("software-raytracer_amd")


def scene_path(name):
    return os.path.join(SCENES, name + ".json")
