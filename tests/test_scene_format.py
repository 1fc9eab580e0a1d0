"""Host C++ scene reader/writer (software-raytracer_amd/host/scene.cpp) against the
reference's format rules (Raytracer/Scene.hpp:27-104, Object.hpp:27-43) and against an
independent python reader."""
import ctypes as C
import json
import math

import numpy as np
import pytest

from conftest import SCENE_NAMES, scene_path


def _bytes(ptr, n, typ):
    return bytes(C.string_at(ptr, C.sizeof(typ) * n))


@pytest.mark.parametrize("name", SCENE_NAMES)
def test_cpp_loader_equals_python_loader(srt, oracle, name):
    s = srt.host.Scene(scene_path(name))
    assert s.error == ""
    ptr, n = s.objects()
    parr, pn = oracle.make_objects(oracle.load_scene_json_py(scene_path(name)))
    assert n == pn and _bytes(ptr, n, srt.Object) == _bytes(parr, pn, oracle.Object)


def test_scene1_matches_its_recorded_generator(srt):
    """Raytracer.cpp:299-325 records how Scene1.json was generated: an 8x8 grid of r=0.2
    spheres at ((i-4)*0.6, -1, (j+10)*0.3), then (0,0,5) r=1, (4,4,8) r=2 E=50,
    (0,-1001.2,5) r=1000.  A known answer for the loader."""
    s = srt.host.Scene(scene_path("Scene1"))
    ptr, n = s.objects()
    assert n == 67
    f = np.float32
    k = 0
    for i in range(8):
        for j in range(8):
            o = ptr[k]
            assert o.type == srt.capi.OBJ_SPHERE and o.radius == f(0.2)
            assert list(o.position) == [f(float(i - 4) * 0.6), f(-1), f((j + 10) * 0.3)]  # float*double, then to float
            assert list(o.material.emissive_color) == [0, 0, 0] and o.material.smoothness == 1.0
            k += 1
    big, emit, ground = ptr[64], ptr[65], ptr[66]
    assert list(big.position) == [0, 0, 5] and big.radius == 1
    assert list(emit.position) == [4, 4, 8] and emit.radius == 2 and list(emit.material.emissive_color) == [50, 50, 50]
    assert list(ground.position) == [0, f(-1001.2), 5] and ground.radius == 1000


def _write(tmp_path, obj, name="s.json"):
    p = tmp_path / name
    p.write_text(obj if isinstance(obj, str) else json.dumps(obj))
    return str(p)


def test_defaults_and_clamps(srt, tmp_path):
    S = {"SceneName": "n", "SceneObjects": [
        # no Material key -> Material() defaults incl. SpecularAmount 0 (Common.hpp:313-318)
        {"Name": "a", "Position": [1, 2, 3], "Renderer": {"Type": "Sphere", "Radius": 2}},
        # empty Material -> loader defaults, SpecularAmount 0.1 (Scene.hpp:61-68)
        {"Name": "b", "Position": [0, 0, 0], "Material": {}, "Renderer": {"Type": "Cube", "Size": [1, 2, 3]}},
        # negative colour components clamp to 0 (Common.hpp:254-257); Metalness is ignored
        {"Name": "c", "Position": [0, 0, 0], "Material": {"Color": [-1, 0.5, 2], "Metalness": 0.9},
         "Renderer": {"Type": "Teapot"}},
    ]}
    s = srt.host.Scene(_write(tmp_path, S))
    assert s.error == "" and len(s) == 3 and s.name == "n"
    ptr, _ = s.objects()
    a, b, c = ptr[0], ptr[1], ptr[2]
    assert (a.type, a.radius, list(a.position)) == (srt.capi.OBJ_SPHERE, 2.0, [1, 2, 3])
    assert (a.material.smoothness, a.material.specular_amount) == (0.5, 0.0)
    assert list(a.material.base_color) == [1, 1, 1] and list(a.material.specular_color) == [1, 1, 1]
    assert b.type == srt.capi.OBJ_BOX and list(b.half_size) == [1, 2, 3]
    assert (b.material.smoothness, b.material.specular_amount) == (0.5, np.float32(0.1))
    assert c.type == srt.capi.OBJ_NONE  # unknown type: inert, still occupies a slot (Scene.hpp:53-55)
    assert list(c.material.base_color) == [0, 0.5, 2] and c.material.specular_amount == np.float32(0.1)
    assert [s.object_name(i) for i in range(3)] == ["a", "b", "c"]


def test_error_behaviour_keeps_objects_loaded_so_far(srt, tmp_path):
    # missing file -> silently empty (Scene.hpp:30-32)
    s = srt.host.Scene(str(tmp_path / "nope.json"))
    assert len(s) == 0 and s.error == ""
    # malformed JSON -> nothing loaded, message kept (:75-77)
    s = srt.host.Scene(_write(tmp_path, '{"SceneName": "", "SceneObjects": [', "bad.json"))
    assert len(s) == 0 and "parse error" in s.error
    # missing SceneName -> null is not a string -> exception before any object (:35)
    s = srt.host.Scene(_write(tmp_path, {"SceneObjects": []}, "noname.json"))
    assert len(s) == 0 and "string" in s.error
    # second object lacks "Name": the first stays, loading stops there (:71,75-77)
    ok = {"Name": "", "Position": [0, 0, 0], "Renderer": {"Type": "Sphere", "Radius": 1}}
    bad = {"Position": [0, 0, 0], "Renderer": {"Type": "Sphere", "Radius": 1}}
    s = srt.host.Scene(_write(tmp_path, {"SceneName": "", "SceneObjects": [ok, bad, ok]}, "partial.json"))
    assert len(s) == 1 and s.error != ""
    # sphere without Radius -> type error -> stop
    s = srt.host.Scene(_write(tmp_path, {"SceneName": "", "SceneObjects": [{"Name": "", "Position": [0, 0, 0], "Renderer": {"Type": "Sphere"}}]}, "norad.json"))
    assert len(s) == 0 and "number" in s.error
    # no SceneObjects key -> empty scene, no error
    s = srt.host.Scene(_write(tmp_path, {"SceneName": "x"}, "noobj.json"))
    assert len(s) == 0 and s.error == ""


@pytest.mark.parametrize("name", SCENE_NAMES)
def test_save_roundtrip(srt, tmp_path, name):
    """Save (dump(4), sorted keys, Grisu2 number spelling) reproduces the reference's own files
    BYTE FOR BYTE — they were written by the reference's Scene::Save (Scene.hpp:88-100) — and a
    reload gives the identical flattened scene."""
    s = srt.host.Scene(scene_path(name))
    assert s.dump() == open(scene_path(name)).read()
    out = str(tmp_path / "out.json")
    s.save_as(out)
    assert open(out, "rb").read() == open(scene_path(name), "rb").read()
    s2 = srt.host.Scene(out)
    p1, n1 = s.objects()
    p2, n2 = s2.objects()
    assert n1 == n2 and _bytes(p1, n1, srt.Object) == _bytes(p2, n2, srt.Object)


def _same_to_float32(a, b):
    if isinstance(a, dict):
        return a.keys() == b.keys() and all(_same_to_float32(a[k], b[k]) for k in a)
    if isinstance(a, list):
        return len(a) == len(b) and all(_same_to_float32(x, y) for x, y in zip(a, b))
    if isinstance(a, float) or isinstance(b, float):
        return np.float32(a) == np.float32(b)
    return a == b


def test_writer_layout(srt, tmp_path):
    s = srt.host.Scene(str(tmp_path / "new.json"), load=False)
    assert s.dump() == '{\n    "SceneName": "",\n    "SceneObjects": []\n}'
    o = srt.Object()
    o.type = srt.capi.OBJ_SPHERE
    o.radius = 0.5
    o.position = (C.c_float * 3)(1, 2, 3)
    o.material.smoothness = 0.5
    o.material.base_color = (C.c_float * 3)(1, 1, 1)
    o.material.specular_color = (C.c_float * 3)(1, 1, 1)
    s.add(o, 'q"x')
    d = json.loads(s.dump())
    e = d["SceneObjects"][0]
    assert e["Name"] == 'q"x' and e["Renderer"] == {"Type": "Sphere", "Radius": 0.5}
    assert e["Material"]["Metalness"] == e["Material"]["SpecularAmount"] == 0.0  # Object.hpp:33,40
    assert list(e.keys()) == sorted(e.keys()) and list(e["Material"].keys()) == sorted(e["Material"].keys())
    assert s.remove(0) and not s.remove(0) and len(s) == 0


@pytest.mark.parametrize("v,text", [
    (1.0, "1.0"), (0.0, "0.0"), (-0.0, "-0.0"), (0.1, "0.1"), (1e-5, "1e-05"), (1e21, "1e+21"),
    (0.20000000298023224, "0.20000000298023224"), (1000.0, "1000.0"), (123456789012345680.0, "1.2345678901234568e+17"),
    (0.0001, "0.0001"), (5e-324, "5e-324"), (float("nan"), "null"), (-1001.2000122070312, "-1001.2000122070313"),
    (1e15, "1e+15"), (1e14, "100000000000000.0"), (1.7976931348623157e308, "1.7976931348623157e+308"),
])
def test_format_double(srt, v, text):
    assert srt.host.format_double(v) == text


def test_writer_numbers_round_trip(srt):
    import random
    import struct

    rnd = random.Random(7)
    for _ in range(20000):
        d = struct.unpack("<d", struct.pack("<Q", rnd.getrandbits(64)))[0]
        if d != d or d in (float("inf"), float("-inf")):
            continue
        assert float(srt.host.format_double(d)) == d
    for _ in range(20000):  # the doubles scenes actually contain: exact images of floats
        f = np.float32(rnd.uniform(-2000, 2000))
        assert np.float32(float(srt.host.format_double(float(f)))) == f


def test_rotate_about_axis_is_rodrigues(srt):
    basis = [1, 0, 0, 0, 1, 0, 0, 0, 1]
    out = srt.host.rotate_about_axis(basis, 0.3, (0, 1, 0))  # Common.hpp:287-291
    c, s_ = math.cos(0.3), math.sin(0.3)
    expect = [c, 0, -s_, 0, 1, 0, s_, 0, c]
    assert np.allclose(out, expect, atol=1e-6)
    assert srt.host.rotate_about_axis(basis, 0.0, (1, 0, 0)) == basis  # Raytracer.cpp:297
