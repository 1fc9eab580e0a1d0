"""Randomized parity sweep: random scenes (extreme coordinates, tiny / huge / negative radii,
cameras inside objects, overlapping boxes, random triangle soups), random cameras and render
parameters.  HIP vs oracle, bit-exact.  Backs the culling-bound proofs with evidence."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _random_scene(oracle, rng, kind):
    objs, meshes = [], []
    # "offset": unit-sized geometry 50 000 units from the origin — coordinates carry only ~8 significant bits
    # of the object sizes, which is where absolute-magnitude rounding would break a non-conservative cull
    scale = {"unit": 1.0, "far": 1.0e4, "tiny": 1.0e-2, "mixed": 1.0, "offset": 1.0}[kind]
    off = np.array({"unit": (0, 0, 6), "far": (3.0e4, -2.0e4, 5.0e4), "tiny": (0, 0, 0.06), "mixed": (0, 0, 6), "offset": (2.5e4, -1.5e4, 4.0e4)}[kind])
    n_sph = int(rng.integers(0, 90))
    for _ in range(n_sph):
        r = float(rng.uniform(0.02, 0.6) * scale)
        if kind == "mixed" and rng.uniform() < 0.1:
            r *= float(rng.choice([30.0, 300.0, -1.0]))  # a few huge ones and a negative radius (r*r is what counts)
        p = off + rng.uniform(-4, 4, 3) * scale
        objs.append(dict(type=oracle.OBJ_SPHERE, position=tuple(float(v) for v in p), radius=r,
                         base=tuple(float(v) for v in rng.uniform(0, 1, 3)), emissive=tuple(float(v) for v in rng.uniform(0, 2, 3) * (rng.uniform() < 0.15)),
                         specular_amount=float(rng.uniform(0, 1)), smoothness=float(rng.uniform(0, 1)), specular=tuple(float(v) for v in rng.uniform(0.5, 1, 3))))
    for _ in range(int(rng.integers(0, 5))):
        p = off + rng.uniform(-4, 4, 3) * scale
        objs.append(dict(type=oracle.OBJ_BOX, position=tuple(float(v) for v in p), half_size=tuple(float(v) for v in rng.uniform(0.1, 3, 3) * scale),
                         base=tuple(float(v) for v in rng.uniform(0, 1, 3)), specular_amount=float(rng.uniform(0, 1)), smoothness=float(rng.uniform(0, 1))))
    if rng.uniform() < 0.5:
        nt = int(rng.integers(1, 200))
        V = (rng.uniform(-1.5, 1.5, (3 * nt, 3)) * scale).astype(np.float32)
        V[1::3] = V[0::3] + (rng.uniform(-0.5, 0.5, (nt, 3)) * scale).astype(np.float32)
        V[2::3] = V[0::3] + (rng.uniform(-0.5, 0.5, (nt, 3)) * scale).astype(np.float32)
        T = np.arange(3 * nt, dtype=np.uint32).reshape(nt, 3)
        if rng.uniform() < 0.25:  # broken input: an index out of range, a non-finite vertex — such triangles never hit
            T[int(rng.integers(0, nt)), int(rng.integers(0, 3))] = 3 * nt + int(rng.integers(0, 1000))
            V[int(rng.integers(0, 3 * nt)), int(rng.integers(0, 3))] = [np.inf, -np.inf, np.nan][int(rng.integers(0, 3))]
        meshes.append((V, T))
        for _ in range(int(rng.integers(1, 3))):
            p = off + rng.uniform(-2, 2, 3) * scale
            objs.append(dict(type=oracle.OBJ_MESH, position=tuple(float(v) for v in p), mesh=0, base=tuple(float(v) for v in rng.uniform(0, 1, 3)),
                             emissive=(0.3, 0.3, 0.3) if rng.uniform() < 0.3 else (0, 0, 0)))
    if rng.uniform() < 0.3:
        objs.insert(int(rng.integers(0, len(objs) + 1)), dict(type=oracle.OBJ_NONE))
    order = rng.permutation(len(objs))
    return [objs[i] for i in order], meshes, off, scale


import os


@pytest.mark.parametrize("seed", range(int(os.environ.get("SRT_FUZZ_N", "24"))))
def test_random_scene_parity(srt, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    kind = ["unit", "far", "tiny", "mixed", "offset"][seed % 5]
    objs, meshes, off, scale = _random_scene(oracle, rng, kind)
    oarr, n = oracle.make_objects(objs)
    marr, mn, keep = oracle.make_meshes(meshes)
    w, h = int(rng.integers(40, 130)), int(rng.integers(30, 90))
    cam = oracle.Camera()
    pos = off - np.array([0, 0, 6.0]) * scale + rng.uniform(-1, 1, 3) * scale * (0 if seed % 3 else 1)
    if seed % 5 == 0 and objs:
        for o in objs:
            if o.get("type") == oracle.OBJ_SPHERE:
                pos = np.array(o["position"]) + 0.3 * abs(o["radius"])  # camera inside a sphere
                break
    basis = srt.host.rotate_about_axis([1, 0, 0, 0, 1, 0, 0, 0, 1], float(rng.uniform(-0.5, 0.5)), (0, 1, 0))
    basis = srt.host.rotate_about_axis(basis, float(rng.uniform(-0.3, 0.3)), basis[0:3])
    cam.position = oracle.f3(pos)
    cam.right, cam.up, cam.forward = oracle.f3(basis[0:3]), oracle.f3(basis[3:6]), oracle.f3(basis[6:9])
    cam.fov_degrees = int(rng.integers(15, 104))
    # every sixth case: enough samples per pixel for the small-tile (>= 16) and sample-chunk (>= 64, meshes >= 32) launches
    spp = int(rng.integers(16, 90)) if seed % 6 == 5 else int(rng.integers(1, 4))
    kw = dict(spp=spp, bounces=int(rng.integers(0, 9)), seed=int(rng.integers(0, 2**31)),
              first_sample=int(rng.integers(1, 50)), reset=bool(rng.integers(0, 2)))
    acc0 = rng.uniform(0, 2, (h, w, 4)).astype(np.float32)
    acc0[..., 3] = 0
    pt = srt.PathTracer(w, h)
    pt.set_meshes(C.cast(marr, C.POINTER(srt.Mesh)), mn)
    pt.set_scene(C.cast(oarr, C.POINTER(srt.Object)), n)
    pt.set_camera(srt.Camera.from_buffer_copy(bytes(cam)))
    pt.write_accumulator(acc0)
    pt.render(count_rays=True, **kw)
    ofb, oacc, orays = oracle.render(oarr, n, oracle.default_environment(), cam, w, h, accumulator=acc0, meshes=(marr, mn) if mn else None, **kw)
    gacc = pt.accumulator()
    assert pt.stats().rays == orays, (kind, n)
    assert np.array_equal(gacc.view(np.uint32), oacc.view(np.uint32)), (kind, n, int((gacc.view(np.uint32) != oacc.view(np.uint32)).any(-1).sum()))
    assert np.array_equal(pt.framebuffer(), ofb)
    pt.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("SRT_FUZZ_BLOCKS_N", "12"))))
def test_random_scene_progressive_blocks(srt, oracle, seed):
    """The same random scenes through progressive-block launches (steps 2..9, random stripe widths, row bands that cut
    blocks, preview shader or path tracing, starting or continuing a frame, one or several samples): block-grid launches and
    the one-lane-per-pixel fallback against the oracle's per-pixel walk."""
    rng = np.random.default_rng(5000 + seed)
    kind = ["unit", "far", "tiny", "mixed", "offset"][seed % 5]
    objs, meshes, off, scale = _random_scene(oracle, rng, kind)
    oarr, n = oracle.make_objects(objs)
    marr, mn, keep = oracle.make_meshes(meshes)
    w, h = int(rng.integers(40, 200)), int(rng.integers(30, 120))
    cam = oracle.Camera()
    pos = off - np.array([0, 0, 6.0]) * scale + rng.uniform(-1, 1, 3) * scale
    basis = srt.host.rotate_about_axis([1, 0, 0, 0, 1, 0, 0, 0, 1], float(rng.uniform(-0.5, 0.5)), (0, 1, 0))
    cam.position = oracle.f3(pos)
    cam.right, cam.up, cam.forward = oracle.f3(basis[0:3]), oracle.f3(basis[3:6]), oracle.f3(basis[6:9])
    cam.fov_degrees = int(rng.integers(15, 104))
    steps = int(rng.integers(2, 10))
    rb = int(rng.integers(0, h - 1))
    re = int(rng.integers(rb + 1, h + 1))
    kw = dict(spp=int(rng.choice([1, 1, 2, 5])), bounces=int(rng.integers(0, 6)), seed=int(rng.integers(0, 2**31)), first_sample=int(rng.integers(1, 50)),
              reset=bool(rng.integers(0, 2)), steps=steps, stripe_width=int(rng.choice([0, w // 16 + 1, int(rng.integers(1, w + 5))])),
              preview=bool(rng.uniform() < 0.3), rows=(rb, re))
    acc0 = rng.uniform(0, 2, (h, w, 4)).astype(np.float32)
    acc0[..., 3] = 0
    pt = srt.PathTracer(w, h)
    pt.set_meshes(C.cast(marr, C.POINTER(srt.Mesh)), mn)
    pt.set_scene(C.cast(oarr, C.POINTER(srt.Object)), n)
    pt.set_camera(srt.Camera.from_buffer_copy(bytes(cam)))
    pt.write_accumulator(acc0)
    pt.render(**kw)
    ofb, oacc, _ = oracle.render(oarr, n, oracle.default_environment(), cam, w, h, accumulator=acc0, meshes=(marr, mn) if mn else None, **kw)
    gacc = pt.accumulator()
    assert np.array_equal(gacc.view(np.uint32), oacc.view(np.uint32)), (kind, n, kw, int((gacc.view(np.uint32) != oacc.view(np.uint32)).any(-1).sum()))
    assert np.array_equal(pt.framebuffer(rows=(rb, re)), ofb[rb:re]), (kind, n, kw)
    pt.close()


@pytest.mark.parametrize("with_mesh", [False, True])
def test_scene_larger_than_lds(srt, oracle, with_mesh):
    """3400 spheres + 60 boxes: the scene image (~220 KB) cannot sit in LDS, so the kernel reads it
    from HBM through the same accessors (SCENE_LDS == false instantiation).  Same bits as the oracle,
    rendering and picking."""
    rng = np.random.default_rng(77)
    objs = []
    for _ in range(3400):
        objs.append(dict(type=oracle.OBJ_SPHERE, position=tuple(float(v) for v in (rng.uniform(-6, 6), rng.uniform(-3, 3), rng.uniform(4, 16))),
                         radius=float(rng.uniform(0.03, 0.25)) * (8.0 if rng.uniform() < 0.01 else 1.0),
                         base=tuple(float(v) for v in rng.uniform(0, 1, 3)), emissive=tuple(float(v) for v in rng.uniform(0, 3, 3) * (rng.uniform() < 0.05)),
                         specular_amount=float(rng.uniform(0, 1)), smoothness=float(rng.uniform(0, 1))))
    for _ in range(60):
        objs.append(dict(type=oracle.OBJ_BOX, position=tuple(float(v) for v in (rng.uniform(-6, 6), rng.uniform(-3, 3), rng.uniform(4, 16))),
                         half_size=tuple(float(v) for v in rng.uniform(0.05, 0.5, 3)), base=tuple(float(v) for v in rng.uniform(0, 1, 3)),
                         specular_amount=float(rng.uniform(0, 1)), smoothness=float(rng.uniform(0, 1))))
    objs.append(dict(type=oracle.OBJ_SPHERE, position=(0.0, -1003.5, 8.0), radius=1000.0, base=(0.6, 0.6, 0.6)))
    meshes = []
    if with_mesh:
        V, T = oracle.uv_sphere(0.8, 12, 16)
        meshes.append((V, T))
        objs.append(dict(type=oracle.OBJ_MESH, position=(0.3, 0.2, 3.0), mesh=0, base=(0.9, 0.5, 0.2), smoothness=0.3))
    order = rng.permutation(len(objs))
    objs = [objs[i] for i in order]
    oarr, n = oracle.make_objects(objs)
    marr, mn, keep = oracle.make_meshes(meshes)
    w, h = 96, 64
    cam = oracle.default_camera()
    pt = srt.PathTracer(w, h)
    pt.set_meshes(C.cast(marr, C.POINTER(srt.Mesh)), mn)
    pt.set_scene(C.cast(oarr, C.POINTER(srt.Object)), n)
    pt.set_camera(srt.Camera.from_buffer_copy(bytes(cam)))
    kw = dict(spp=3, bounces=5, seed=11)
    pt.render(count_rays=True, **kw)
    ofb, oacc, orays = oracle.render(oarr, n, oracle.default_environment(), cam, w, h, meshes=(marr, mn) if mn else None, **kw)
    assert pt.stats().rays == orays
    assert np.array_equal(pt.accumulator().view(np.uint32), oacc.view(np.uint32))
    assert np.array_equal(pt.framebuffer(), ofb)
    if not with_mesh:  # the oracle's closest-object probe is analytic only
        d, nn, pp, t = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)(), C.c_float()
        origin = (C.c_float * 3)(0, 0, 0)
        for (x, y) in [(0, 0), (48, 32), (20, 10), (90, 60), (48, 5), (33, 33)]:
            oracle.lib().srt_oracle_ray_direction(C.byref(cam), w, h, x, y, d)
            assert pt.pick(x, y) == oracle.lib().srt_oracle_closest(oarr, n, origin, d, nn, pp, C.byref(t))
    pt.close()
