"""HIP path vs the CPU oracle, through the C-ABI, on the same seeded inputs.

Bar: bit-exact framebuffer AND float accumulator against the oracle with the shared powf
(include/srt_defs.h); within 1 LSB per 8-bit channel against the oracle with libm powf
(north_star: "within 1 ULP per channel after tone-map").
"""
import ctypes as C

import numpy as np
import pytest

from conftest import SCENE_NAMES, scene_path

pytestmark = pytest.mark.gpu


def _setup(srt, oracle, name, w, h):
    objs = oracle.load_scene_json_py(scene_path(name))
    oarr, n = oracle.make_objects(objs)
    garr = C.cast(oarr, C.POINTER(srt.Object))  # same POD layout (include/srt_pathtrace.h)
    pt = srt.PathTracer(w, h)
    pt.set_scene(garr, n)
    pt.set_camera(srt.default_camera())
    return pt, oarr, n


def _channels(fb):
    return np.stack([(fb >> 24) & 255, (fb >> 16) & 255, (fb >> 8) & 255, fb & 255], -1).astype(np.int32)


@pytest.mark.parametrize("name", SCENE_NAMES)
@pytest.mark.parametrize("w,h,spp,bounces", [(256, 256, 1, 4), (160, 90, 4, 8), (64, 48, 3, 0), (97, 61, 2, 2)])
def test_frame_bit_exact(srt, oracle, name, w, h, spp, bounces):
    pt, oarr, n = _setup(srt, oracle, name, w, h)
    pt.render(spp=spp, bounces=bounces, seed=0, count_rays=True)
    fb, acc, st = pt.framebuffer(), pt.accumulator(), pt.stats()
    ofb, oacc, orays = oracle.render(oarr, n, oracle.default_environment(), oracle.default_camera(), w, h,
                                     spp=spp, bounces=bounces, seed=0, pow_mode=oracle.POW_SHARED)
    assert st.path_samples == w * h * spp
    assert st.rays == orays
    assert np.array_equal(acc.view(np.uint32), oacc.view(np.uint32)), \
        "accumulator differs in %d floats" % int((acc.view(np.uint32) != oacc.view(np.uint32)).sum())
    assert np.array_equal(fb, ofb)
    # reference-faithful libm powf: at most 1 LSB per channel
    lfb, _, _ = oracle.render(oarr, n, oracle.default_environment(), oracle.default_camera(), w, h,
                              spp=spp, bounces=bounces, seed=0, pow_mode=oracle.POW_LIBM)
    assert np.abs(_channels(fb) - _channels(lfb)).max() <= 1
    pt.close()


@pytest.mark.parametrize("nsph,nbox", [(20, 0), (300, 3), (1500, 0)])
def test_random_large_scenes(srt, oracle, nsph, nbox):
    """Random scenes: few spheres (no clustering), hundreds (clusters of 4 plus boxes), 1500 (clusters
    grow to K = 24; LDS image close to 100 KB).  Bit-exact vs the oracle."""
    rng = np.random.default_rng(nsph)
    objs = [dict(type=oracle.OBJ_SPHERE, position=(0, -1001, 5), radius=1000, base=(.8, .8, .8))]
    for _ in range(nsph):
        objs.append(dict(type=oracle.OBJ_SPHERE, position=(float(rng.uniform(-6, 6)), float(rng.uniform(-0.8, 3)), float(rng.uniform(3, 14))),
                         radius=float(rng.uniform(0.05, 0.25)), base=tuple(float(v) for v in rng.uniform(0.2, 1, 3)),
                         emissive=(2.0, 1.5, 1.0) if rng.uniform() < 0.05 else (0, 0, 0),
                         specular_amount=float(rng.uniform(0, 1)), smoothness=float(rng.uniform(0, 1))))
    for _ in range(nbox):
        objs.insert(int(rng.integers(0, len(objs))), dict(type=oracle.OBJ_BOX, position=(float(rng.uniform(-4, 4)), float(rng.uniform(0, 2)), float(rng.uniform(5, 10))),
                                                          half_size=tuple(float(v) for v in rng.uniform(0.2, 0.8, 3)), base=(.3, .6, .9)))
    oarr, n = oracle.make_objects(objs)
    w, h = 112, 63
    pt = srt.PathTracer(w, h)
    pt.set_scene(C.cast(oarr, C.POINTER(srt.Object)), n)
    pt.set_camera(srt.default_camera())
    pt.render(spp=2, bounces=5, seed=1, count_rays=True)
    ofb, oacc, orays = oracle.render(oarr, n, oracle.default_environment(), oracle.default_camera(), w, h, spp=2, bounces=5, seed=1)
    assert pt.stats().rays == orays
    assert np.array_equal(pt.accumulator().view(np.uint32), oacc.view(np.uint32))
    assert np.array_equal(pt.framebuffer(), ofb)
    pt.close()


@pytest.mark.parametrize("case", ["inf half size", "nan centre", "inf camera", "finite", "huge box 3e30", "huge box 2e38", "huge centre 5e33",
                                  "camera at 1e30", "camera at 3e38", "box just under the bound"])
def test_boxes_with_non_finite_numbers(srt, oracle, case):
    """The box slab test has a NaN-free form (v_max3 / v_min3) that the kernel may only take when the scene's boxes and every ray
    that counts are finite; anything else goes through the comparisons as the reference writes them.  Scenes and cameras on
    both sides of that decision, bit-exact vs the oracle."""
    inf, nan = float("inf"), float("nan")
    objs = [dict(type=oracle.OBJ_SPHERE, position=(0, -1001, 5), radius=1000, base=(.8, .8, .8)),
            dict(type=oracle.OBJ_BOX, position=(0.5, 0.5, 6), half_size=(0.7, 0.5, 0.6), base=(.3, .6, .9), smoothness=0.8, specular_amount=0.5),
            dict(type=oracle.OBJ_BOX, position=(-1.5, 0.2, 5), half_size=(0.4, 0.4, 0.4), base=(.9, .4, .3)),
            dict(type=oracle.OBJ_SPHERE, position=(1.5, 0.3, 4), radius=0.5, base=(.5, .9, .5), emissive=(1, 1, 1))]
    if case == "inf half size":
        objs.append(dict(type=oracle.OBJ_BOX, position=(3, 1, 8), half_size=(inf, 0.5, 0.5), base=(.5, .5, .5)))
    if case == "nan centre":
        objs.append(dict(type=oracle.OBJ_BOX, position=(nan, 1, 8), half_size=(0.5, 0.5, 0.5), base=(.5, .5, .5)))
    # round 4 (advice): FINITE is not enough for the NaN-free form — o - c or a slab product can overflow to infinity on the way
    # (slopes reach 1e8) and -inf + inf is a NaN the reference's comparisons propagate and v_max3 / v_min3 drop.  The kernel now
    # asks for magnitudes below 1e29 (boxes: srt_set_scene; origins: per wave); these cases lie on both sides of that bound.
    if case == "huge box 3e30":
        objs.append(dict(type=oracle.OBJ_BOX, position=(0, 0, 9), half_size=(3e30, 3.4e30, 3e30), base=(.5, .5, .5)))
    if case == "huge box 2e38":
        objs.append(dict(type=oracle.OBJ_BOX, position=(1, 2, 9), half_size=(2e38, 0.5, 2e38), base=(.5, .7, .5)))
    if case == "huge centre 5e33":
        objs.append(dict(type=oracle.OBJ_BOX, position=(5e33, -5e33, 5e33), half_size=(6e33, 6e33, 6e33), base=(.6, .5, .5)))
    if case == "box just under the bound":
        objs.append(dict(type=oracle.OBJ_BOX, position=(9e28, 0, 0), half_size=(9.5e28, 9e28, 9e28), base=(.6, .5, .7)))
    oarr, n = oracle.make_objects(objs)
    cam, ocam = srt.default_camera(), oracle.default_camera()
    if case == "inf camera":
        cam.position = (C.c_float * 3)(0.0, inf, 0.0)
        ocam.position = (C.c_float * 3)(0.0, inf, 0.0)
    if case == "camera at 1e30":
        cam.position = (C.c_float * 3)(1e30, 0.5, -1e30)
        ocam.position = (C.c_float * 3)(1e30, 0.5, -1e30)
    if case == "camera at 3e38":
        cam.position = (C.c_float * 3)(0.0, 3e38, 0.0)
        ocam.position = (C.c_float * 3)(0.0, 3e38, 0.0)
    w, h = 96, 54
    pt = srt.PathTracer(w, h)
    pt.set_scene(C.cast(oarr, C.POINTER(srt.Object)), n)
    pt.set_camera(cam)
    pt.render(spp=4, bounces=6, seed=2, count_rays=True)
    ofb, oacc, orays = oracle.render(oarr, n, oracle.default_environment(), ocam, w, h, spp=4, bounces=6, seed=2)
    assert pt.stats().rays == orays
    assert np.array_equal(pt.accumulator().view(np.uint32), oacc.view(np.uint32))
    assert np.array_equal(pt.framebuffer(), ofb)
    pt.close()


def test_object_count_limit(srt, oracle):
    """The hit key holds the list index in 15 bits: 32768 objects are refused with a message, 4000 (an
    image beyond LDS, served from HBM — see test_scene_larger_than_lds) are accepted."""
    pt = srt.PathTracer(16, 16)
    objs = [dict(type=oracle.OBJ_SPHERE, position=(i * 0.01, 0, 5), radius=0.01) for i in range(4000)]
    oarr, n = oracle.make_objects(objs)
    pt.set_scene(C.cast(oarr, C.POINTER(srt.Object)), n)
    objs = [dict(type=oracle.OBJ_NONE)] * 32768
    oarr, n = oracle.make_objects(objs)
    with pytest.raises(srt.SrtError) as e:
        pt.set_scene(C.cast(oarr, C.POINTER(srt.Object)), n)
    assert e.value.code == srt.capi.ERR_INVALID_ARG and "32767" in str(e.value)
    # a failed srt_set_scene leaves the context without a scene, not with a half-replaced one
    pt.set_camera(srt.default_camera())
    with pytest.raises(srt.SrtError) as e:
        pt.render(spp=1, bounces=1, seed=0)
    assert e.value.code == srt.capi.ERR_STATE
    pt.close()


def test_colours_that_stress_the_clamps(srt, oracle):
    """Round 3 removed every clamp0 of the reference that is provably the identity (kernel header: a clamp changes strictly negative
    values only; sums and products of values that are never strictly negative stay so) and clamps the material / environment
    colours once, when the scene image is built.  The oracle still executes every clamp of Color's constructor
    (Common.hpp:253-262).  Negative, -0.0, large and tiny colour components — in materials, in the environment and in an
    accumulator handed in by the caller — must give the same bits.  (No NaN / infinity here: x86 and gfx950 give NaNs different
    payloads, which the byte-level comparison of float accumulators would report although both are "the" NaN.)"""
    objs = oracle.load_scene_json_py(scene_path("Scene_indirect"))
    nasty = [(-1.0, 0.5, 2.0), (-0.0, 0.0, 1.0), (1e6, 1e-30, -1e6), (3.0, 0.25, -0.0), (1e-38, 1.0, -3.0), (0.0, -0.0, 7.5)]
    for i, o in enumerate(objs):
        if o.get("type") in (oracle.OBJ_SPHERE, oracle.OBJ_BOX):
            o["base"] = nasty[i % len(nasty)]
            o["emissive"] = nasty[(i + 2) % len(nasty)] if i % 3 == 0 else o.get("emissive", (0, 0, 0))
            o["specular"] = nasty[(i + 4) % len(nasty)] if i % 4 == 0 else o.get("specular", (1, 1, 1))
    oarr, n = oracle.make_objects(objs)
    w, h = 96, 64
    env = srt.default_environment()
    env.sky_color = (C.c_float * 3)(-2.0, 3.5, 10.0)
    env.ground_color = (C.c_float * 3)(0.08, -0.0, 0.03)
    env.sun_color = (C.c_float * 3)(500.0, -500.0, 1e6)
    oenv = oracle.Environment.from_buffer_copy(bytes(env))
    pt = srt.PathTracer(w, h)
    pt.set_environment(env)
    pt.set_scene(C.cast(oarr, C.POINTER(srt.Object)), n)
    pt.set_camera(srt.default_camera())
    rng = np.random.default_rng(3)
    acc0 = rng.uniform(-2.0, 5.0, (h, w, 4)).astype(np.float32)  # a caller's accumulator with negative entries
    acc0[::7, ::5] = -0.0
    pt.write_accumulator(acc0)
    pt.render(spp=3, bounces=6, seed=5, first_sample=4, reset=False)
    ofb, oacc, _ = oracle.render(oarr, n, oenv, oracle.default_camera(), w, h, spp=3, bounces=6, seed=5, first_sample=4, reset=False, accumulator=acc0)
    assert np.array_equal(pt.accumulator().view(np.uint32), oacc.view(np.uint32)), int((pt.accumulator().view(np.uint32) != oacc.view(np.uint32)).sum())
    assert np.array_equal(pt.framebuffer(), ofb)
    pt.render(spp=40, bounces=8, seed=6)  # sample-chunk-free full launch from a reset, many samples
    ofb, oacc, _ = oracle.render(oarr, n, oenv, oracle.default_camera(), w, h, spp=40, bounces=8, seed=6)
    assert np.array_equal(pt.accumulator().view(np.uint32), oacc.view(np.uint32)) and np.array_equal(pt.framebuffer(), ofb)
    pt.close()


@pytest.mark.parametrize("case", ["radius 3e-12", "radius 0", "radius 1e-30", "radius inf", "radius nan", "radius 2e-11 (inside)", "radius 1e19 (inside)"])
def test_spheres_outside_the_short_square_roots_window(srt, oracle, case):
    """The LDS instantiations take the sphere test's square root in its short form (sqrt_window, csrc/srt_kernel.hip.h), valid
    when every sphere's r*r lies in [2^-72, FLT_MAX]; srt_set_scene sends a scene with a radius outside — tiny, zero, infinite,
    NaN — to the instantiations that keep the library sqrtf.  Bit-exact either way; which way it went shows in the work
    counts (the memory instantiations keep none)."""
    r = {"radius 3e-12": 3e-12, "radius 0": 0.0, "radius 1e-30": 1e-30, "radius inf": float("inf"), "radius nan": float("nan"),
         "radius 2e-11 (inside)": 2e-11, "radius 1e19 (inside)": 1e19}[case]
    objs = oracle.load_scene_json_py(scene_path("Scene1"))
    # the odd sphere sits right in front of the camera, on the axis of the centre pixel's ray, so that rays do meet its centre
    odd = dict(type=oracle.OBJ_SPHERE, position=(0.0, 0.0, 2.0) if r < 1e18 else (0.0, 0.0, 2e19), radius=r, base=(.9, .2, .1), emissive=(0.5, 0.5, 0.5))
    objs.insert(3, odd)
    objs.append(dict(type=oracle.OBJ_SPHERE, position=(0.3, 0.1, 3.0), radius=r, base=(.1, .9, .1)))
    oarr, n = oracle.make_objects(objs)
    w, h = 129, 65
    pt = srt.PathTracer(w, h)
    pt.set_scene(C.cast(oarr, C.POINTER(srt.Object)), n)
    pt.set_camera(srt.default_camera())
    pt.render(spp=3, bounces=4, seed=11, count_rays=True, count_work=True)
    st = pt.stats()
    ofb, oacc, orays = oracle.render(oarr, n, oracle.default_environment(), oracle.default_camera(), w, h, spp=3, bounces=4, seed=11)
    assert st.rays == orays
    # An infinite radius makes every first hit lie at -inf and every bounce ray a NaN: whole colours turn into NaNs — the same
    # floats in the same places, but a NaN's sign and payload are the processor's (x86 and gfx950 differ), so NaNs compare as NaNs
    # here.  KNOWN DEVIATION: the reference's Color carries an alpha through its arithmetic, 0 everywhere unless an
    # environment lerp runs on a NaN parameter, which makes it 0 * NaN; the kernel keeps the accumulator's alpha at 0.  The
    # framebuffer is the same either way (alpha tone-maps to byte 0 from 0 and from NaN).
    acc = pt.accumulator()
    nan = np.isnan(acc)
    assert np.array_equal(nan[..., :3], np.isnan(oacc)[..., :3])
    assert np.array_equal(np.where(nan, 0, acc.view(np.uint32))[..., :3], np.where(nan, 0, oacc.view(np.uint32))[..., :3])
    alpha_same = acc.view(np.uint32)[..., 3] == oacc.view(np.uint32)[..., 3]
    assert np.all(alpha_same | (np.isnan(oacc[..., 3]) & (acc[..., 3] == 0) & np.isnan(oacc[..., :3]).all(-1)))
    if case != "radius inf":
        assert alpha_same.all()
    assert np.array_equal(pt.framebuffer(), ofb)
    assert pt.work_counts().as_dict()["valid"] == (1 if "inside" in case else 0)
    pt.close()
