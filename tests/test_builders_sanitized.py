"""Host-side builders (scene image, 8-wide BVH, JSON / Scene code) under ASan + UBSan on random and
degenerate inputs (NaN / inf / huge coordinates, zero-area and duplicate triangles, out-of-range indices,
mutated JSON), plus a structural check of the BVH: every triangle is referenced by exactly one leaf.
CPU builds only — GPU sanitizers are not available on this pool."""
import os
import shutil
import subprocess

import pytest

import json_corpus
from conftest import scene_path

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SAN = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]

pytestmark = pytest.mark.skipif(shutil.which("g++") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime.h"),
                                reason="needs g++ and the HIP headers")


def _build(tmp_path, src, extra):
    exe = str(tmp_path / "check")
    subprocess.run(["g++"] + SAN + extra + [os.path.join(HERE, "native", src), "-o", exe], check=True, capture_output=True)
    return exe


def test_scene_image_and_bvh_builders(tmp_path):
    exe = _build(tmp_path, "builders_check.cpp", ["-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "software-raytracer_amd", "csrc"),
                                                  "-I" + os.path.join(ROOT, "include")])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout[-400:] + r.stderr[-2000:]


def test_launch_shape_rule(tmp_path):
    """csrc/srt_launch_shape.h — tile height, sample chunks, taper, the fill simulation — is a pure function of its inputs (round 4):
    the shapes BASELINE's configs take on the GPU, reproduced on the CPU under ASan + UBSan."""
    exe = _build(tmp_path, "shape_check.cpp", ["-I" + os.path.join(ROOT, "software-raytracer_amd", "csrc")])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout[-600:] + r.stderr[-2000:]


def test_json_and_scene_code(tmp_path):
    host = os.path.join(ROOT, "software-raytracer_amd", "host")
    exe = _build(tmp_path, "json_check.cpp", ["-I" + host, "-I" + os.path.join(ROOT, "include"), os.path.join(host, "scene.cpp")])
    docs = json_corpus.documents(8000, seed=4242)
    r = subprocess.run([exe] + [scene_path(n) for n in ("Scene1", "Scene_indirect", "Scene3")], input=b"\n".join(docs) + b"\n", capture_output=True)
    assert r.returncode == 0 and r.stdout.startswith(b"ok "), r.stdout[-400:] + r.stderr[-2000:]
