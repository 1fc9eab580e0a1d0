"""Hand-derived known answers for the oracle's restatement of the reference functions.

The reference ships no tests or vectors, so these are analytic cases worked out from the
reference source by hand (file:line cited per case).  They pin the oracle's *semantics*
(including the reference's quirks); bit-level GPU parity is tested in test_gpu_parity.py.
"""
import ctypes as C
import math

import numpy as np
import pytest


def f3(v):
    return (C.c_float * 3)(*v)


def sphere(oracle, c, r, **kw):
    arr, _ = oracle.make_objects([dict(type=oracle.OBJ_SPHERE, position=c, radius=r, **kw)])
    return arr


def box(oracle, c, h, **kw):
    arr, _ = oracle.make_objects([dict(type=oracle.OBJ_BOX, position=c, half_size=h, **kw)])
    return arr


def intersect(oracle, obj, o, d):
    n, p, t = f3((0, 0, 0)), f3((0, 0, 0)), C.c_float(0)
    valid = oracle.lib().srt_oracle_intersect(obj, f3(o), f3(d), n, p, C.byref(t))
    return valid, list(n), list(p), t.value


# ---- Sphere::line_sphere_intersection (Object.hpp:104-141) ---------------------------
def test_sphere_head_on(oracle):
    v, n, p, t = intersect(oracle, sphere(oracle, (0, 0, 5), 1), (0, 0, 0), (0, 0, 1))
    assert v == 1 and t == 4.0 and p == [0, 0, 4] and n == [0, 0, -1]


def test_sphere_behind_is_mirrored_but_misses_when_origin_outside(oracle):
    # tc = abs(dot) (:118-119) mirrors the sphere in front; |L|^2 + 3 tc^2 > r^2 -> miss
    v, *_ = intersect(oracle, sphere(oracle, (0, 0, -5), 1), (0, 0, 0), (0, 0, 1))
    assert v == 0


def test_origin_inside_sphere_gives_negative_distance(oracle):
    # c=(0,0,.5) r=1: tc=.5, d2=0, t1 = .5 - 1 = -.5: the hit is BEHIND the origin (:131-137)
    v, n, p, t = intersect(oracle, sphere(oracle, (0, 0, 0.5), 1), (0, 0, 0), (0, 0, 1))
    assert v == 1 and t == -0.5 and p == [0, 0, -0.5] and n == [0, 0, -1]


def test_sphere_behind_origin_inside_hits_through_abs(oracle):
    # c=(0,0,-.5) r=1: dot=-.5 -> tc=.5, q=(0,0,.5), d2 = 1 == r^2 (not >) -> hit, t1 = .5
    v, n, p, t = intersect(oracle, sphere(oracle, (0, 0, -0.5), 1), (0, 0, 0), (0, 0, 1))
    assert v == 1 and t == 0.5 and p == [0, 0, 0.5] and n == [0, 0, 1]


def test_sphere_tangent_miss(oracle):
    v, *_ = intersect(oracle, sphere(oracle, (1.5, 0, 5), 1), (0, 0, 0), (0, 0, 1))
    assert v == 0


# ---- Box::iBox (Object.hpp:173-200) ----------------------------------------------------
def test_box_axis_aligned_ray_degenerates_to_a_miss(oracle):
    # sign(0) = 0 (Common.hpp:328-333) -> m = 0, t2.x = t2.y = 0 -> tF <= 0 -> miss: the
    # one-pixel cross the reference draws at x = W/2, y = H/2 in box scenes
    v, *_ = intersect(oracle, box(oracle, (0, 0, 5), (1, 1, 1)), (0, 0, 0), (0, 0, 1))
    assert v == 0


def test_box_diagonal_hit_corner(oracle):
    s = 1 / math.sqrt(3)
    v, n, p, t = intersect(oracle, box(oracle, (0, 0, 0), (1, 1, 1)), (-5, -5, -5), (s, s, s))
    assert v == 1
    assert t == pytest.approx(4 * math.sqrt(3), rel=1e-6)
    assert p == pytest.approx([-1, -1, -1], abs=1e-5)
    # entry-face normal -sign(rd)*step*step with all three t1 equal or nearly so: components in {0,-1}
    assert set(n) <= {0.0, -0.0, -1.0} and -1.0 in n


def test_box_face_hit_and_normal(oracle):
    d = np.array([0.1, 0.2, 1.0]) / np.linalg.norm([0.1, 0.2, 1.0])
    v, n, p, t = intersect(oracle, box(oracle, (0, 0, 5), (1, 1, 1)), (0, 0, 0), tuple(d))
    assert v == 1 and n == [0, 0, -1]
    assert p[2] == pytest.approx(4.0, abs=1e-5) and t == pytest.approx(4.0 / d[2], rel=1e-6)


def test_box_inside_returns_far_distance_with_entry_normal(oracle):
    # origin inside: tN < 0.01 -> tF accepted, normal still computed from t1 (entry face) (:192-195)
    d = np.array([0.1, 0.2, 1.0]) / np.linalg.norm([0.1, 0.2, 1.0])
    v, n, p, t = intersect(oracle, box(oracle, (0, 0, 0), (1, 1, 1)), (0, 0, 0), tuple(d))
    assert v == 1 and t == pytest.approx(1.0 / d[2], rel=1e-6) and n == [0, 0, -1]


def test_box_range_limits(oracle):
    d = np.array([0.1, 0.2, 1.0]) / np.linalg.norm([0.1, 0.2, 1.0])
    # beyond distBound.y = 10000 (:226)
    v, *_ = intersect(oracle, box(oracle, (1200, 2400, 12000), (1, 1, 1)), (0, 0, 0), tuple(d))
    assert v == 0


# ---- GetClosestObject (Raytracer.cpp:123-140) ------------------------------------------
def closest(oracle, objs, o, d):
    arr, n = oracle.make_objects(objs)
    nn, p, t = f3((0, 0, 0)), f3((0, 0, 0)), C.c_float(0)
    idx = oracle.lib().srt_oracle_closest(arr, n, f3(o), f3(d), nn, p, C.byref(t))
    return idx, t.value


def test_closest_picks_nearest_and_first_on_ties(oracle):
    S = oracle.OBJ_SPHERE
    objs = [dict(type=S, position=(0, 0, 9), radius=1), dict(type=S, position=(0, 0, 5), radius=1),
            dict(type=S, position=(0, 0, 5), radius=1)]
    assert closest(oracle, objs, (0, 0, 0), (0, 0, 1)) == (1, 4.0)  # strict <: index 1 beats its twin 2


def test_closest_negative_distance_wins(oracle):
    S = oracle.OBJ_SPHERE
    objs = [dict(type=S, position=(0, 0, 5), radius=1), dict(type=S, position=(0, 0, 0.5), radius=1)]
    assert closest(oracle, objs, (0, 0, 0), (0, 0, 1)) == (1, -0.5)  # Raytracer.cpp:132 has no t > 0 test


def test_closest_miss_and_inert_objects(oracle):
    objs = [dict(type=oracle.OBJ_NONE, position=(0, 0, 5)), dict(type=oracle.OBJ_SPHERE, position=(9, 9, 5), radius=1)]
    idx, t = closest(oracle, objs, (0, 0, 0), (0, 0, 1))
    assert idx == -1 and math.isinf(t)


# ---- GetEnvironmentColor (Raytracer.cpp:77-89) -----------------------------------------
def env(oracle, d, mode=None):
    e = oracle.default_environment()
    out = f3((0, 0, 0))
    oracle.lib().srt_oracle_environment(C.byref(e), f3(d), oracle.POW_LIBM if mode is None else mode, out)
    return list(out)


def test_environment_default_constants(oracle):
    e = oracle.default_environment()
    s = np.float32(1) / np.sqrt(np.float32(3))
    assert list(e.sun_direction) == [s, -s, -s]
    assert list(e.sky_color) == [2.0, 3.5, 10.0] and list(e.horizon_color) == [5.0, 4.5, 2.5]
    assert list(e.ground_color) == [np.float32(.08), np.float32(.06), np.float32(.03)]
    assert list(e.sun_color) == [500.0] * 3


def test_environment_zenith_nadir_sun(oracle):
    f = np.float32
    # straight up: powf(1,.1)=1 -> t = Sky; then Lerp(t, Sky*0.1, 1) = Sky*0.1
    assert env(oracle, (0, 1, 0)) == [f(2) * f(0.1), f(3.5) * f(0.1), f(10) * f(0.1)]
    # straight down: Lerp(Horizon, Ground, 1) = Ground
    assert env(oracle, (0, -1, 0)) == [f(.08), f(.06), f(.03)]
    # into the sun: + (500,500,500)
    s = float(f(1) / np.sqrt(f(3)))
    up = env(oracle, (-s, s, s))
    nosun = env(oracle, (s, s, -s))
    assert all(a > 500 for a in up) and all(b < 11 for b in nosun)
    # both powf variants agree here to 1 ulp
    a, b = env(oracle, (0.3, 0.5, 0.8), oracle.POW_LIBM), env(oracle, (0.3, 0.5, 0.8), oracle.POW_SHARED)
    assert np.allclose(a, b, rtol=3e-7, atol=0)


# ---- GetRayDirection (Raytracer.cpp:106-122) -------------------------------------------
def test_ray_direction_centre_and_corner(oracle):
    cam = oracle.default_camera()
    out = f3((0, 0, 0))
    oracle.lib().srt_oracle_ray_direction(C.byref(cam), 256, 256, 128, 128, out)
    assert list(out) == [0, 0, 1]  # nX = nY = 0: pixel CORNER sampling, no +0.5 (:109-110)
    oracle.lib().srt_oracle_ray_direction(C.byref(cam), 1920, 1080, 0, 0, out)
    d = np.array(list(out), dtype=np.float64)
    ld = math.tan(math.radians(55) / 2)  # vertical FOV (:112-115)
    ref = np.array([-ld * 1920 / 1080, -ld, 1.0])
    ref /= np.linalg.norm(ref)
    assert np.allclose(d, ref, atol=2e-6)
    assert abs(np.linalg.norm(d) - 1) < 1e-6


# ---- SetScreenPixel tone-map + pack (Raytracer.cpp:73-75, Common.hpp:189-208) -----------
@pytest.mark.parametrize("rgba,expect", [
    ((0, 0, 0, 0), 0x00000000),              # alpha = 0/0 = NaN -> (int)NaN = INT_MIN on x86 -> byte 0
    ((1, 1, 1, 0), 0x007F7F7F),              # 1/(1+1) * 255 = 127.5 -> truncation
    ((3, 1, 0, 0), 0x00BF7F00),              # 0.75*255 = 191.25 -> 0xBF
    ((50, 50, 50, 0), 0x00FAFAFA),           # 50/51*255 = 250
    ((float("nan"), 0.5, float("inf"), 0), 0x00005500),  # NaN and inf/inf -> byte 0; .5/1.5*255 = 85
    ((1, 1, 1, 2), 0xFF7F7F7F),              # a/(0+a) = 1 -> 255
])
def test_tonemap_pack(oracle, rgba, expect):
    arr = (C.c_float * 4)(*rgba)
    assert oracle.lib().srt_oracle_tonemap_pack(arr) == expect


# ---- RaytraceScene sample sequence (Raytracer.cpp:141-185) -----------
def test_draw_counts_and_ray_counts(oracle):
    S = oracle.OBJ_SPHERE
    cam, e = oracle.default_camera(), oracle.default_environment()
    rgba, rays, draws = (C.c_float * 4)(), C.c_uint32(), C.c_uint32()

    def trace(objs, px, py, bounces):
        arr, n = oracle.make_objects(objs)
        oracle.lib().srt_oracle_trace_sample(arr, n, C.byref(e), C.byref(cam), 64, 64, px, py, 1, bounces, 0,
                                             oracle.POW_LIBM, rgba, C.byref(rays), C.byref(draws))
        return list(rgba), rays.value, draws.value

    # primary miss: env colour, 1 ray, 0 draws
    c, r, d = trace([], 32, 32, 8)
    assert (r, d) == (1, 0) and c[3] == 0 and c[:3] == env(oracle, (0, 0, 1))
    # primary hit, MAXBOUNCES = 0: colour = emissive, 1 ray, 1 draw (the coin at :165)
    objs = [dict(type=S, position=(0, 0, 5), radius=1, emissive=(3, 2, 1), base=(.5, .5, .5))]
    c, r, d = trace(objs, 32, 32, 0)
    assert (r, d) == (1, 1) and c == [3, 2, 1, 0]
    # camera inside a big sphere: abs(tc) makes the primary ray hit the sphere BEHIND the camera
    # (negative distance), the normal there points outward, so the bounce starts outside and
    # escapes: 2 rays, draws = coin + 3 (no second coin on a miss, :178-181)
    objs = [dict(type=S, position=(0, 0, 0), radius=50, emissive=(1, 1, 1), base=(1, 1, 1), specular_amount=0.0)]
    for B in (1, 3, 8):
        c, r, d = trace(objs, 10, 50, B)
        assert (r, d) == (2, 4)
        assert all(v > 1.0 for v in c[:3])  # L = E + env * base
    # invariant on a real scene: b = rays-1 bounces ran; draws = 1 + 3b + hits,
    # hits = b when the path ended on the bounce limit, b-1 when it escaped
    from conftest import scene_path
    sc = oracle.load_scene_json_py(scene_path("Scene_indirect"))
    seen = set()
    for px in range(3, 64, 6):
        for py in range(2, 64, 7):
            c, r, d = trace(sc, px, py, 8)
            b = r - 1
            assert 1 <= r <= 9
            assert d in (0, 1 + 4 * b, 4 * b) or (b == 0 and d == 0)
            seen.add(r)
    assert len(seen) >= 4  # paths of several lengths were exercised


# ---- preview shader (SIMPLEDRAW, Raytracer.cpp:147-160) and progressive blocks (:233-248) ----
def test_preview_shader_known_answers(oracle):
    S = oracle.OBJ_SPHERE
    # head-on hit of a matte sphere (k = 0): colour = base + emissive, no env term; alpha 0
    objs = [dict(type=S, position=(0, 0, 5), radius=1, base=(.2, .4, .6), emissive=(.1, 0, 0), specular_amount=0.0)]
    arr, n = oracle.make_objects(objs)
    kw = dict(spp=1, bounces=8, seed=0, preview=True)
    fb, acc, rays = oracle.render(arr, n, oracle.default_environment(), oracle.default_camera(), 64, 64, **kw)
    f = np.float32
    assert list(acc[32, 32]) == [f(.2) * f(1) + f(.1), f(.4), f(.6), 0]
    assert rays == 64 * 64  # one ray per pixel, no bounces
    # selected: fresnel = smoothstep(0, .5, 1 - dot(-n, d)); head-on gives 0 -> unchanged; at the
    # silhouette it saturates to 1 -> Color(3, 3, 0)
    fb2, acc2, _ = oracle.render(arr, n, oracle.default_environment(), oracle.default_camera(), 64, 64, selected=0, **kw)
    assert list(acc2[32, 32]) == list(acc[32, 32])
    ys, xs = np.nonzero((acc2[..., 0] == 3) & (acc2[..., 1] == 3) & (acc2[..., 2] == 0))
    assert len(xs) > 10 and not np.array_equal(acc2, acc)
    # a mirror (k = s = 1, base 0) looking straight up reflects env(reflect(d, n))
    objs = [dict(type=S, position=(0, -101, 5), radius=100, base=(0, 0, 0), specular_amount=1.0, smoothness=1.0)]
    arr, n = oracle.make_objects(objs)
    _, acc3, _ = oracle.render(arr, n, oracle.default_environment(), oracle.default_camera(), 64, 64, **kw)
    assert (acc3[5, 32, :3] > 0.5).all()  # floor pixel shows sky


def test_block_anchor_formula_equals_the_literal_loop_nest(oracle):
    """steps x steps replication: the per-pixel anchor formula used by the oracle's normal
    walk (and by the GPU kernel) equals renderArea's literal loops over 16 column stripes."""
    from conftest import scene_path
    arr, n = oracle.make_objects(oracle.load_scene_json_py(scene_path("Scene2")))
    env, cam = oracle.default_environment(), oracle.default_camera()
    for w, h, steps in [(160, 90, 2), (131, 67, 4), (97, 50, 8)]:
        div = w // 16 + 1  # Raytracer.cpp:330
        for preview in (True, False):
            kw = dict(spp=1, bounces=3, seed=1, steps=steps, stripe_width=div, preview=preview)
            lit = oracle.render(arr, n, env, cam, w, h, threads=16, split=oracle.SPLIT_REF_COLS, **kw)
            fm = oracle.render(arr, n, env, cam, w, h, threads=3, split=oracle.SPLIT_ROWS, **kw)
            assert np.array_equal(lit[0], fm[0]) and np.array_equal(lit[1], fm[1])
            if not preview:
                assert lit[2] < fm[2]  # the literal walk traces one ray per block, the formula one per pixel
            # blocks really replicate: pixel (1,1) equals its anchor (0,0) when steps > 1
            assert np.array_equal(fm[1][0, 0], fm[1][1, 1])


# ---- EXTENSION: triangle test (this project's definition, srt_pathtrace.h) -------------------
def test_triangle_known_answers(oracle):
    L = oracle.lib()
    L.srt_oracle_triangle.argtypes = [C.POINTER(C.c_float)] * 7 + [C.POINTER(C.c_float)]
    L.srt_oracle_triangle.restype = C.c_int

    def tri(v0, v1, v2, o, d):
        n, p, t = f3((0, 0, 0)), f3((0, 0, 0)), C.c_float(0)
        ok = L.srt_oracle_triangle(f3(v0), f3(v1), f3(v2), f3(o), f3(d), n, p, C.byref(t))
        return ok, list(n), list(p), t.value

    A, B, Cc = (-1, -1, 4), (1, -1, 4), (0, 1, 4)
    ok, n, p, t = tri(A, B, Cc, (0, 0, 0), (0, 0, 1))
    assert ok == 1 and t == 4.0 and p == [0, 0, 4] and n == [0, 0, -1]        # normal turned against the ray
    ok, n, p, t = tri(A, Cc, B, (0, 0, 0), (0, 0, 1))
    assert ok == 1 and n == [0, 0, -1]                                        # winding does not matter (two-sided)
    assert tri(A, B, Cc, (0, 0, 0), (0, 0, -1))[0] == 0                       # behind the origin
    assert tri(A, B, Cc, (3, 0, 0), (0, 0, 1))[0] == 0                        # outside
    assert tri(A, B, Cc, (0, 0, 3.995), (0, 0, 1))[0] == 0                    # t < 0.01 (the Box bound, Object.hpp:226)
    assert tri(A, B, Cc, (0, 0, 0), (1, 0, 0))[0] == 0                        # parallel
    ok, n, p, t = tri(A, B, Cc, (0, -1, 0), (0, 0, 1))
    assert ok == 1 and t == 4.0                                               # on the edge v = 0 counts as a hit


def test_uv_sphere_generator(oracle):
    V, T = oracle.uv_sphere(1.0, 224, 224)
    assert T.shape == (99904, 3) and V.shape == (2 + 223 * 224, 3)
    r = np.linalg.norm(V.astype(np.float64), axis=1)
    assert np.abs(r - 1).max() < 1e-6
    # outward winding: normals point away from the centre
    a, b, c = V[T[:, 0]].astype(np.float64), V[T[:, 1]].astype(np.float64), V[T[:, 2]].astype(np.float64)
    nrm = np.cross(b - a, c - a)
    assert (np.einsum("ij,ij->i", nrm, (a + b + c) / 3) > 0).all()
