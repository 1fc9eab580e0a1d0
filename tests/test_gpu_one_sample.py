"""One-sample launches — the reference's own frame loop adds ONE sample per frame (Raytracer.cpp:572-595).  They run through
pathtrace_kernel like every other launch shape (two dedicated one-sample kernels were built and dropped in round 3, DESIGN.md
§4.9; these tests were written for them and stay as the coverage of the frame loop's launch shape): odd sizes, 0 bounces, the
sample-by-sample frame loop against one multi-sample launch, row bands onto an accumulated frame, meshes, a scene image beyond
LDS.  Bit-exact against the oracle: framebuffer, accumulator, ray count."""
import ctypes as C

import numpy as np
import pytest

from conftest import SCENE_NAMES, scene_path

pytestmark = pytest.mark.gpu


def _setup(srt, oracle, name, w, h):
    oarr, n = oracle.make_objects(oracle.load_scene_json_py(scene_path(name)))
    pt = srt.PathTracer(w, h)
    pt.set_scene(C.cast(oarr, C.POINTER(srt.Object)), n)
    pt.set_camera(srt.default_camera())
    return pt, oarr, n


def _same(pt, ofb, oacc, rows=None):
    acc, fb = pt.accumulator(), pt.framebuffer()
    if rows is not None:
        h = acc.shape[0]
        ys = slice(h - rows[1], h - rows[0])
        return np.array_equal(acc[ys].view(np.uint32), oacc[ys].view(np.uint32)) and np.array_equal(fb[rows[0]:rows[1]], ofb[rows[0]:rows[1]])
    return np.array_equal(acc.view(np.uint32), oacc.view(np.uint32)) and np.array_equal(fb, ofb)


@pytest.mark.parametrize("name", SCENE_NAMES)
@pytest.mark.parametrize("w,h,bounces", [(320, 180, 8), (131, 77, 3), (33, 9, 16), (640, 40, 0)])
def test_one_sample_frame_bit_exact(srt, oracle, name, w, h, bounces):
    """odd sizes: tiles cut at the right edge and at the band's last rows; 0 bounces: the emissive colour only"""
    pt, oarr, n = _setup(srt, oracle, name, w, h)
    pt.render(spp=1, bounces=bounces, seed=3, count_rays=True)
    ofb, oacc, orays = oracle.render(oarr, n, oracle.default_environment(), oracle.default_camera(), w, h, spp=1, bounces=bounces, seed=3)
    assert pt.stats().rays == orays and pt.stats().path_samples == w * h
    assert _same(pt, ofb, oacc)
    pt.close()


@pytest.mark.parametrize("name", ["Scene1", "Scene_indirect", "Scene3"])
def test_frame_loop_one_sample_per_launch(srt, oracle, name):
    """the reference's loop: frame f adds sample f (reset at f = 1).  Twelve one-sample launches == one 12-sample launch of the
    pool kernel == the oracle; then a camera move restarts the accumulation."""
    w, h = 200, 120
    pt, oarr, n = _setup(srt, oracle, name, w, h)
    for f in range(1, 13):
        pt.render(spp=1, bounces=8, seed=0, first_sample=f, reset=(f == 1))
    ofb, oacc, _ = oracle.render(oarr, n, oracle.default_environment(), oracle.default_camera(), w, h, spp=12, bounces=8, seed=0)
    assert _same(pt, ofb, oacc)
    acc12 = pt.accumulator()
    pt.render(spp=12, bounces=8, seed=0)  # the pool kernel, all 12 at once
    assert np.array_equal(pt.accumulator().view(np.uint32), acc12.view(np.uint32))
    cam = srt.default_camera()
    cam.position = (C.c_float * 3)(0.4, 0.3, -1.0)
    pt.set_camera(cam)
    ocam = oracle.default_camera()
    ocam.position = (C.c_float * 3)(0.4, 0.3, -1.0)
    for f in range(1, 4):
        pt.render(spp=1, bounces=8, seed=0, first_sample=f, reset=(f == 1))
    ofb, oacc, _ = oracle.render(oarr, n, oracle.default_environment(), ocam, w, h, spp=3, bounces=8, seed=0)
    assert _same(pt, ofb, oacc)
    pt.close()


def test_one_sample_row_bands_and_resume(srt, oracle):
    """a rank's band as its own one-sample launch: only its rows change; bands in any order give the frame; a sample added to
    an accumulated frame (first_sample = 7, no reset) reads the running mean back"""
    w, h = 256, 144
    pt, oarr, n = _setup(srt, oracle, "Scene_indirect", w, h)
    env, cam = oracle.default_environment(), oracle.default_camera()
    pt.render(spp=6, bounces=5, seed=9)
    _, oacc6, _ = oracle.render(oarr, n, env, cam, w, h, spp=6, bounces=5, seed=9)
    for rows in ((100, 144), (0, 37), (37, 100)):
        pt.render(spp=1, bounces=5, seed=9, first_sample=7, reset=False, rows=rows)
    ofb, oacc, _ = oracle.render(oarr, n, env, cam, w, h, spp=1, bounces=5, seed=9, first_sample=7, reset=False, accumulator=oacc6)
    assert _same(pt, ofb, oacc)
    # a band alone: the other rows keep what they had
    before = pt.accumulator().copy()
    pt.render(spp=1, bounces=5, seed=9, first_sample=8, reset=False, rows=(40, 48))
    after = pt.accumulator()
    assert np.array_equal(after[: h - 48].view(np.uint32), before[: h - 48].view(np.uint32)) and np.array_equal(after[h - 40:].view(np.uint32), before[h - 40:].view(np.uint32))
    assert not np.array_equal(after[h - 48: h - 40].view(np.uint32), before[h - 48: h - 40].view(np.uint32))
    pt.close()


def test_one_sample_with_mesh(srt, oracle):
    """EXTENSION: one-sample launches of a mesh scene (rays that wait for a mesh phase are parked)"""
    objs = oracle.load_scene_json_py(scene_path("Scene1"))
    big = objs[64]
    objs[64] = dict(type=oracle.OBJ_MESH, position=big["position"], mesh=0, base=big["base"], emissive=big["emissive"],
                    smoothness=big["smoothness"], specular_amount=big["specular_amount"], specular=big["specular"])
    V, T = oracle.uv_sphere(1.0, 24, 32)
    oarr, n = oracle.make_objects(objs)
    marr, mn, keep = oracle.make_meshes([(V, T)])
    w, h = 240, 135
    pt = srt.PathTracer(w, h)
    pt.set_meshes(C.cast(marr, C.POINTER(srt.Mesh)), mn)
    pt.set_scene(C.cast(oarr, C.POINTER(srt.Object)), n)
    pt.set_camera(srt.default_camera())
    for f in range(1, 4):
        pt.render(spp=1, bounces=8, seed=2, first_sample=f, reset=(f == 1), count_rays=True)
        rays_last = pt.stats().rays
    ofb, oacc, _ = oracle.render(oarr, n, oracle.default_environment(), oracle.default_camera(), w, h, spp=3, bounces=8, seed=2, meshes=(marr, mn))
    assert _same(pt, ofb, oacc)
    _, _, orays3 = oracle.render(oarr, n, oracle.default_environment(), oracle.default_camera(), w, h, spp=1, bounces=8, seed=2, first_sample=3, reset=False,
                                 accumulator=np.zeros((h, w, 4), np.float32), meshes=(marr, mn))
    assert rays_last == orays3
    pt.close()


def test_one_sample_scene_beyond_lds(srt, oracle):
    """a scene image that does not fit into LDS (the HBM-scene instantiation)"""
    rng = np.random.default_rng(5)
    objs = [dict(type=oracle.OBJ_SPHERE, position=(0, -1001, 5), radius=1000, base=(.8, .8, .8))]
    for _ in range(3400):
        objs.append(dict(type=oracle.OBJ_SPHERE, position=(float(rng.uniform(-6, 6)), float(rng.uniform(-0.8, 3)), float(rng.uniform(3, 14))),
                         radius=float(rng.uniform(0.03, 0.12)), base=tuple(float(v) for v in rng.uniform(0.2, 1, 3))))
    oarr, n = oracle.make_objects(objs)
    w, h = 96, 54
    pt = srt.PathTracer(w, h)
    pt.set_scene(C.cast(oarr, C.POINTER(srt.Object)), n)
    pt.set_camera(srt.default_camera())
    pt.render(spp=1, bounces=4, seed=1, count_rays=True)
    ofb, oacc, orays = oracle.render(oarr, n, oracle.default_environment(), oracle.default_camera(), w, h, spp=1, bounces=4, seed=1)
    assert pt.stats().rays == orays
    assert _same(pt, ofb, oacc)
    pt.close()
