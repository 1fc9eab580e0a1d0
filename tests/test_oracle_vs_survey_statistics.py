"""The only numbers that come from the REAL reference: path statistics the survey measured by running the
unmodified Raytracer.cpp hot path — mean GetClosestObject calls per path-sample
for the six shipped scenes and for Scene1 at 2 / 4 / 16 bounces, and mean rand() draws per path-sample.

They were taken with another random generator than this project's (the reference's own is unreproducible,
DESIGN.md §3), so they pin the oracle statistically, not bit for bit: a restatement that got the path loop wrong
(a bounce too many, the escape test, the hemisphere flip, which rays count, negative distances, the box bounds)
moves these means by far more than the tolerance.  1280x720, camera at origin, FOV 55, 1 spp = 921,600 paths:
the sampling error of a mean of ~2-8 rays is ~0.2 %; the published figures carry three digits, so the bar is
1 % of the figure plus half a unit of its last printed digit.
"""
import ctypes as C

import pytest

from conftest import scene_path

W, H = 1280, 720

# rays per path-sample at 8 bounces
RBAR_8 = {
    "Scene1": 2.09,
    "Scene1_reflection": 2.20,
    "Scene2": 2.13,
    "Scene3": 2.64,
    "Scene3_indirect": 7.29,
    "Scene_indirect": 7.95,
}
# Scene1 at other bounce limits
RBAR_SCENE1 = {2: 1.77, 4: 1.94, 16: 2.21}
# rand() draws per path-sample at 8 bounces
DRAWS_8 = {"Scene1": 4.39, "Scene_indirect": 28.5}


def _tol(published):
    digits = len(repr(published).split(".")[1])
    return 0.01 * published + 0.5 * 10.0 ** -digits


def _scene(oracle, name):
    return oracle.make_objects(oracle.load_scene_json_py(scene_path(name)))


def _rbar(oracle, name, bounces):
    arr, n = _scene(oracle, name)
    # libm powf, as the reference's source says (the environment colour is terminal: no statistic depends on it)
    _, _, rays = oracle.render(arr, n, oracle.default_environment(), oracle.default_camera(55), W, H, spp=1, bounces=bounces, seed=0,
                               pow_mode=oracle.POW_LIBM)
    return rays / float(W * H)


@pytest.mark.parametrize("name", sorted(RBAR_8))
def test_rays_per_sample_8_bounces(oracle, name):
    got = _rbar(oracle, name, 8)
    assert abs(got - RBAR_8[name]) <= _tol(RBAR_8[name]), "%s: oracle %.4f rays/sample, the survey measured %.2f on the reference" % (name, got, RBAR_8[name])


@pytest.mark.parametrize("bounces", sorted(RBAR_SCENE1))
def test_rays_per_sample_scene1_other_bounce_limits(oracle, bounces):
    got = _rbar(oracle, "Scene1", bounces)
    assert abs(got - RBAR_SCENE1[bounces]) <= _tol(RBAR_SCENE1[bounces]), "Scene1 at %d bounces: oracle %.4f, survey %.2f" % (bounces, got, RBAR_SCENE1[bounces])


@pytest.mark.parametrize("name", sorted(DRAWS_8))
def test_random_draws_per_sample(oracle, name):
    """Draws = 0 for a primary miss, else 1 + 3 per bounce executed + 1 per secondary hit.
    Counted by the oracle's per-sample probe on every 4th pixel of every 4th row (57,600 paths; sampling error ~1 %
    of the mean for Scene1's long-tailed distribution, hence the wider bar: 2.5 %)."""
    arr, n = _scene(oracle, name)
    env, cam = oracle.default_environment(), oracle.default_camera(55)
    L = oracle.lib()
    rgba = (C.c_float * 4)()
    rays, draws = C.c_uint32(0), C.c_uint32(0)
    tot_d = tot_r = cnt = 0
    for y in range(2, H, 4):
        for x in range(2, W, 4):
            L.srt_oracle_trace_sample(arr, n, C.byref(env), C.byref(cam), W, H, x, y, 1, 8, 0, oracle.POW_LIBM, rgba, C.byref(rays), C.byref(draws))
            tot_d += draws.value
            tot_r += rays.value
            cnt += 1
    got = tot_d / float(cnt)
    assert abs(got - DRAWS_8[name]) <= 0.025 * DRAWS_8[name] + 0.05, "%s: oracle %.3f draws/sample, survey %.3g" % (name, got, DRAWS_8[name])
    # the same probe's rays agree with the frame-wide figure too
    assert abs(tot_r / float(cnt) - RBAR_8[name]) <= 0.025 * RBAR_8[name]
