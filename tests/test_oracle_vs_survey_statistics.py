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


# share of the primary rays that hit an object (measured on the unmodified reference)
PRIMARY_HIT = {"Scene1": (0.56, 0.006), "Scene_indirect": (0.999, 0.0006)}  # published figure, half a unit of its last digit + 1 %


@pytest.mark.parametrize("name", sorted(PRIMARY_HIT))
def test_share_of_primary_rays_that_hit(oracle, name):
    """No random numbers involved: with ONE bounce every traced pixel casts exactly one more ray, so rays = pixels + primary hits.
    "56 % of primaries hit" (Scene1) / "99.9 %" (Scene_indirect) are the survey's figures from the real reference."""
    arr, n = _scene(oracle, name)
    _, _, rays = oracle.render(arr, n, oracle.default_environment(), oracle.default_camera(55), W, H, spp=1, bounces=1, seed=0, pow_mode=oracle.POW_LIBM)
    share = rays / float(W * H) - 1.0
    published, half_digit = PRIMARY_HIT[name]
    assert abs(share - published) <= 0.01 * published + half_digit, "%s: oracle %.4f of the primaries hit, the survey saw %.3g" % (name, share, published)


def test_degenerate_ray_cross_in_box_scenes(oracle):
    """SURVEY §7 'hard parts', seen in the survey's render of the real reference: sign() returns 0 for a zero component
    (Common.hpp:328-333), which degenerates iBox's slab test (Object.hpp:175-186: m = 0, so t1 = t2 = 0 on that axis and tF <= 0
    rejects) — a ray whose direction has an exactly zero x or y component misses EVERY box.  With the camera at the origin those
    are the primary rays of column W / 2 and of row H / 2, so Box scenes show a one-pixel cross there.  Known answer on the oracle:
    in Scene_indirect's closed room the neighbours of the cross see walls (boxes), the cross itself sees what lies behind."""
    import ctypes as C

    arr, n = _scene(oracle, "Scene_indirect")
    objs = oracle.load_scene_json_py(scene_path("Scene_indirect"))
    is_box = [o["type"] == oracle.OBJ_BOX for o in objs]
    cam = oracle.default_camera(55)
    L = oracle.lib()
    f3 = C.c_float * 3

    def primary(x, y):
        d, nn, p, t = f3(), f3(), f3(), C.c_float()
        L.srt_oracle_ray_direction(C.byref(cam), W, H, x, y, d)
        return L.srt_oracle_closest(arr, n, f3(0, 0, 0), d, nn, p, C.byref(t)), tuple(d)

    col_boxes = off_boxes = row_boxes = 0
    for y in range(0, H, 7):
        idx, d = primary(W // 2, y)
        assert d[0] == 0.0                                     # nX = (W/2)/W*2 - 1 = 0 exactly (Raytracer.cpp:109)
        col_boxes += idx >= 0 and is_box[idx]
        for dx in (-1, 1):
            idx, d = primary(W // 2 + dx, y)
            assert d[0] != 0.0
            off_boxes += idx >= 0 and is_box[idx]
    for x in range(0, W, 7):
        idx, d = primary(x, H // 2)
        assert d[1] == 0.0
        row_boxes += idx >= 0 and is_box[idx]
    assert col_boxes == 0 and row_boxes == 0                   # the cross: no primary ray on it ever reports a box
    assert off_boxes >= 0.25 * 2 * len(range(0, H, 7))         # one pixel beside it, the rays that pass the sphere grid end on the room's walls (30 % of the rows)
    nbr_boxes = 0
    for x in range(0, W, 7):
        idx, _ = primary(x, H // 2 + 1)
        nbr_boxes += idx >= 0 and is_box[idx]
    assert nbr_boxes >= 0.7 * len(range(0, W, 7))              # ... and one row beside the horizontal bar most rays do (78 %)
    # ... and it is visible in the frame: the centre column differs from both neighbours wherever they see a wall
    fb, _, _ = oracle.render(arr, n, oracle.default_environment(), cam, W, H, spp=1, bounces=0, seed=0, pow_mode=oracle.POW_LIBM,
                             cols=(W // 2 - 1, W // 2 + 2))
    col = fb[:, W // 2 - 1:W // 2 + 2]
    differs = ((col[:, 1] != col[:, 0]) & (col[:, 1] != col[:, 2])).mean()
    assert differs > 0.2
