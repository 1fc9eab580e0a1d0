"""The native (C++ / C-ABI, no Python in the data path) multi-GPU row-stripe renderer of SURVEY §8e:
host/renderer.hpp MultiGpuRenderer = one srt_context and stream per device, memory-row bands (of equal estimated cost by default,
of equal height on request), joined by srt_gather_band (device-to-device copies into the first device's framebuffer).  On a one-GPU box the N contexts
all live on device 0 — the control flow, the band arithmetic and the stream ordering are the same."""
import ctypes as C
import subprocess
import os

import numpy as np
import pytest

from conftest import ROOT, scene_path

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("equal", [True, False])
@pytest.mark.parametrize("name,n_parts,w,h,spp", [("Scene_indirect", 3, 200, 113, 3), ("Scene1", 8, 1920, 1080, 4), ("Scene3", 5, 333, 97, 20)])
def test_multi_renderer_equals_single_device_and_oracle(srt, oracle, name, n_parts, w, h, spp, equal):
    scene = srt.host.Scene(scene_path(name))
    m = srt.host.MultiRenderer([0] * n_parts, w, h)
    m.set_scene(scene)
    m.configure(fov=55, max_bounces=6, seed=3)
    if equal:
        m.use_equal_bands()
    else:
        m.set_auto_balance_min_samples(0)  # (the automatic split probes only for requests of >= 32 samples per device; these are smaller)
    m.render_samples(spp, count_rays=True)
    bands = [m.band(i) for i in range(n_parts)]
    assert bands[0][0] == 0 and bands[-1][1] == h and all(bands[i][1] == bands[i + 1][0] for i in range(n_parts - 1))
    if equal:
        assert max(b - a for a, b in bands) - min(b - a for a, b in bands) <= 1  # equal bands (first h % N one row taller)
    elif name == "Scene1":
        assert bands[0][1] - bands[0][0] > 2 * (bands[-1][1] - bands[-1][0])  # the default split: the sky band is far taller than the floor band
    fb = m.framebuffer()
    st = m.stats()
    assert sum(s.path_samples for s in st) == w * h * spp
    # single device, same settings
    objs, n = scene.objects_copy()
    pt = srt.PathTracer(w, h)
    pt.set_scene(objs, n)
    pt.set_camera(srt.default_camera())
    pt.render(spp=spp, bounces=6, seed=3, count_rays=True)
    assert np.array_equal(fb, pt.framebuffer())
    assert sum(s.rays for s in st) == pt.stats().rays
    if w * h * spp <= 2_000_000:
        ofb, _, _ = oracle.render(C.cast(objs, C.POINTER(oracle.Object)), n, oracle.default_environment(), oracle.default_camera(), w, h,
                                  spp=spp, bounces=6, seed=3)
        assert np.array_equal(fb, ofb)
    # a second batch continues the running means on every band (resume), like one device
    m.render_samples(2)
    pt.render(spp=2, bounces=6, seed=3, first_sample=spp + 1, reset=False)
    assert np.array_equal(m.framebuffer(), pt.framebuffer())
    m.close()
    pt.close()


def test_balanced_bands_equal_the_single_device_frame(srt):
    """MultiGpuRenderer::BalanceBands: bands of equal ESTIMATED cost (srt_estimate_row_costs) — the frame must not change,
    the bands must, and the estimate must be the same every time (all ranks of a multi-process job rely on that)."""
    w, h, n_parts = 640, 360, 4
    scene = srt.host.Scene(scene_path("Scene1"))
    objs, n = scene.objects_copy()
    pt = srt.PathTracer(w, h)
    pt.set_scene(objs, n)
    pt.set_camera(srt.default_camera())
    c1, c2 = pt.estimate_row_costs(8, 0), pt.estimate_row_costs(8, 0)
    assert c1 == c2 and len(c1) == h and min(c1) > 0
    assert sum(c1[h // 2:]) > 3 * sum(c1[:h // 2])  # Scene1: the bottom half (floor, grid) costs far more than the sky
    pt.render(spp=4, bounces=8, seed=0)
    m = srt.host.MultiRenderer([0] * n_parts, w, h)
    m.set_scene(scene)
    m.configure(fov=55, max_bounces=8, seed=0)
    equal = [m.band(i) for i in range(n_parts)]  # (before the first render: the constructor's equal bands)
    m.balance_bands()
    bands = [m.band(i) for i in range(n_parts)]
    assert bands != equal and bands[0][0] == 0 and bands[-1][1] == h and all(bands[i][1] == bands[i + 1][0] for i in range(n_parts - 1))
    assert bands[0][1] - bands[0][0] > bands[-1][1] - bands[-1][0]  # the sky band is the tallest
    m.render_samples(4)
    assert np.array_equal(m.framebuffer(), pt.framebuffer())
    assert [m.band(i) for i in range(n_parts)] == bands  # the default split of render_samples IS balance_bands(): deterministic
    m.use_equal_bands()
    m.render_samples(4)
    assert [m.band(i) for i in range(n_parts)] == equal and np.array_equal(m.framebuffer(), pt.framebuffer())
    m.close()
    pt.close()


def test_automatic_split_is_gated_on_the_work_it_balances(srt):
    """Round 4 (advice): the balance probe costs about 8 sample-frames on ONE device, synchronously.  The automatic split pays it
    only for a request of at least 32 samples per device, reuses a split made for the same scene / camera / bounces (a new seed
    does not probe again), and leaves bands alone that the caller set by hand when told so."""
    w, h, n_parts = 640, 360, 4
    scene = srt.host.Scene(scene_path("Scene1"))
    objs, n = scene.objects_copy()
    pt = srt.PathTracer(w, h)
    pt.set_scene(objs, n)
    pt.set_camera(srt.default_camera())
    m = srt.host.MultiRenderer([0] * n_parts, w, h)
    m.set_scene(scene)
    m.configure(fov=55, max_bounces=8, seed=0)
    equal = [m.band(i) for i in range(n_parts)]
    m.render_samples(4)                                   # 4 < 32 x 4: no probe, the equal bands stay
    assert [m.band(i) for i in range(n_parts)] == equal
    pt.render(spp=4, bounces=8, seed=0)
    assert np.array_equal(m.framebuffer(), pt.framebuffer())
    m.configure(fov=55, max_bounces=8, seed=0)            # the accumulation restarts ...
    m.render_samples(128)                                 # ... with a request worth the probe: balanced bands
    bands = [m.band(i) for i in range(n_parts)]
    assert bands != equal and bands[0][1] - bands[0][0] > bands[-1][1] - bands[-1][0]
    pt.render(spp=128, bounces=8, seed=0)
    assert np.array_equal(m.framebuffer(), pt.framebuffer())
    m.configure(fov=55, max_bounces=8, seed=7)            # only the seed changed: the split is reused, also for a small request
    m.render_samples(2)
    assert [m.band(i) for i in range(n_parts)] == bands
    pt.render(spp=2, bounces=8, seed=7)
    assert np.array_equal(m.framebuffer(), pt.framebuffer())
    m.configure(fov=55, max_bounces=3, seed=7)            # other bounces and a small request: not worth a probe, and the old split
    m.render_samples(2)                                   # is worth no more than equal bands
    assert [m.band(i) for i in range(n_parts)] == equal
    # bands set by hand are left alone under use_manual_bands (and REPLACED without it, as documented)
    hand = [(0, 100), (100, 180), (180, 300), (300, 360)]
    m.use_manual_bands()
    for i, (a, b) in enumerate(hand):
        m.set_row_band(i, a, b)
    m.configure(fov=55, max_bounces=8, seed=0)
    m.render_samples(256)
    assert [m.band(i) for i in range(n_parts)] == hand
    pt.render(spp=256, bounces=8, seed=0)
    assert np.array_equal(m.framebuffer(), pt.framebuffer())
    m.close()
    pt.close()


def test_gather_reports_the_way_it_went(srt):
    """srt_gather_path: on a one-GPU box every gather is a same-device copy; the text is what the first multi-GPU run will
    report about peer access (the cross-device ways have never run here)."""
    w, h = 256, 128
    scene = srt.host.Scene(scene_path("Scene1"))
    objs, n = scene.objects_copy()
    a, b = srt.PathTracer(w, h), srt.PathTracer(w, h)
    assert a.gather_path() == "no gather yet"
    for t in (a, b):
        t.set_scene(objs, n)
        t.set_camera(srt.default_camera())
    a.render(spp=2, bounces=4, seed=0, rows=(0, 64))
    b.render(spp=2, bounces=4, seed=0, rows=(64, 128))
    a.gather_band_from(b, (64, 128))
    assert "same device" in b.gather_path() and a.gather_path() == "no gather yet"
    ref = srt.PathTracer(w, h)
    ref.set_scene(objs, n)
    ref.set_camera(srt.default_camera())
    ref.render(spp=2, bounces=4, seed=0)
    assert np.array_equal(a.framebuffer(), ref.framebuffer())
    for t in (a, b, ref):
        t.close()


def test_balance_probe_leaves_the_frame_alone(srt, oracle):
    """srt_estimate_row_costs runs the PROBE instantiation of the path-trace kernel (the real pool, loop trips counted): it
    must not touch accumulator, framebuffer, ray count or the accumulation state — a frame rendered around a probe equals
    the frame rendered without one — and must work for every instantiation: analytic, mesh, scene image beyond LDS."""
    import ctypes as C
    w, h = 320, 180
    objs = oracle.load_scene_json_py(scene_path("Scene1"))
    big = objs[64]
    mesh_objs = list(objs)
    mesh_objs[64] = dict(type=oracle.OBJ_MESH, position=big["position"], mesh=0, base=big["base"], emissive=big["emissive"],
                         smoothness=big["smoothness"], specular_amount=big["specular_amount"], specular=big["specular"])
    V, T = oracle.uv_sphere(1.0, 24, 32)
    marr, mn, keep = oracle.make_meshes([(V, T)])
    rng = np.random.default_rng(5)
    many = [dict(type=oracle.OBJ_SPHERE, position=(0, -1001, 5), radius=1000, base=(.8, .8, .8))]
    for _ in range(3400):
        many.append(dict(type=oracle.OBJ_SPHERE, position=(float(rng.uniform(-6, 6)), float(rng.uniform(-0.8, 3)), float(rng.uniform(3, 14))),
                         radius=float(rng.uniform(0.03, 0.12)), base=tuple(float(v) for v in rng.uniform(0.2, 1, 3))))
    for name, scene_objs, meshes in (("analytic", objs, None), ("mesh", mesh_objs, (marr, mn)), ("beyond LDS", many, None)):
        oarr, n = oracle.make_objects(scene_objs)
        pt = srt.PathTracer(w, h)
        if meshes:
            pt.set_meshes(C.cast(meshes[0], C.POINTER(srt.Mesh)), meshes[1])
        pt.set_scene(C.cast(oarr, C.POINTER(srt.Object)), n)
        pt.set_camera(srt.default_camera())
        pt.render(spp=3, bounces=6, seed=1, count_rays=True)
        acc, fb, rays = pt.accumulator().copy(), pt.framebuffer().copy(), pt.stats().rays
        c1 = pt.estimate_row_costs(6, 1)
        assert len(c1) == h and min(c1) > 0 and c1 == pt.estimate_row_costs(6, 1), name
        assert np.array_equal(pt.accumulator().view(np.uint32), acc.view(np.uint32)) and np.array_equal(pt.framebuffer(), fb), name
        assert pt.stats().rays == rays, name
        pt.render(spp=2, bounces=6, seed=1, first_sample=4, reset=False)  # the accumulation goes on as if nothing had happened
        ref = srt.PathTracer(w, h)
        if meshes:
            ref.set_meshes(C.cast(meshes[0], C.POINTER(srt.Mesh)), meshes[1])
        ref.set_scene(C.cast(oarr, C.POINTER(srt.Object)), n)
        ref.set_camera(srt.default_camera())
        ref.render(spp=5, bounces=6, seed=1)
        assert np.array_equal(pt.accumulator().view(np.uint32), ref.accumulator().view(np.uint32)), name
        pt.close()
        ref.close()


def test_gather_band_argument_checks(srt):
    a, b, c = srt.PathTracer(64, 32), srt.PathTracer(64, 32), srt.PathTracer(32, 32)
    with pytest.raises(srt.SrtError):
        a.gather_band_from(c, (0, 8))   # different size
    with pytest.raises(srt.SrtError):
        a.gather_band_from(b, (8, 8))   # empty band
    with pytest.raises(srt.SrtError):
        a.gather_band_from(b, (0, 33))  # beyond the frame
    a.gather_band_from(a, (0, 32))      # self: nothing to do
    for t in (a, b, c):
        t.close()


def test_cli_devices_flag(tmp_path):
    """srt_render --devices 0,0,0 writes the same PPM as the single-device run."""
    exe = os.path.join(ROOT, "software-raytracer_amd", "srt_render")
    one, three = tmp_path / "one.ppm", tmp_path / "three.ppm"
    common = [exe, "--scene", scene_path("Scene2"), "--width", "160", "--height", "90", "--spp", "4", "--bounces", "4"]
    r1 = subprocess.run(common + ["--out", str(one)], capture_output=True, text=True, timeout=120)
    r3 = subprocess.run(common + ["--devices", "0,0,0", "--out", str(three)], capture_output=True, text=True, timeout=120)
    assert r1.returncode == 0 and r3.returncode == 0, r1.stderr + r3.stderr
    assert one.read_bytes() == three.read_bytes()
    assert "over 3 parts" in r3.stderr
