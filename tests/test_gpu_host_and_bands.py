"""GPU tests of everything around the kernel: row bands, resume, external output buffers,
the C++ host renderer's frame sequences, full-size (BASELINE) properties."""
import ctypes as C

import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu


def _pt(srt, name, w, h):
    s = srt.host.Scene(scene_path(name))
    objs, n = s.objects_copy()
    pt = srt.PathTracer(w, h)
    pt.set_scene(objs, n)
    pt.set_camera(srt.default_camera())
    return pt, objs, n


def _oracle_frame(oracle, objs, n, w, h, **kw):
    return oracle.render(C.cast(objs, C.POINTER(oracle.Object)), n, oracle.default_environment(), oracle.default_camera(), w, h, **kw)


def test_row_bands_concatenate_to_the_full_frame(srt):
    """Any partition into memory-row bands reproduces the single-launch frame byte for byte
    (RNG keyed by absolute pixel) — the multi-GPU contract, exercised on one GPU."""
    w, h = 200, 117
    pt, objs, n = _pt(srt, "Scene_indirect", w, h)
    pt.render(spp=3, bounces=8, seed=5)
    full_fb, full_acc = pt.framebuffer(), pt.accumulator()
    pt2 = srt.PathTracer(w, h)
    pt2.set_scene(objs, n)
    pt2.set_camera(srt.default_camera())
    for rb, re in [(0, 1), (1, 16), (16, 17), (17, 100), (100, 117)]:  # ragged, not multiples of the tile
        pt2.render(spp=3, bounces=8, seed=5, rows=(rb, re))
        assert np.array_equal(pt2.framebuffer(rows=(rb, re)), full_fb[rb:re])
    assert np.array_equal(pt2.framebuffer(), full_fb)
    assert np.array_equal(pt2.accumulator().view(np.uint32), full_acc.view(np.uint32))


def test_resume_equals_one_shot_and_oracle(srt, oracle):
    w, h = 96, 64
    pt, objs, n = _pt(srt, "Scene2", w, h)
    pt.render(spp=7, bounces=4, seed=3)
    one = (pt.framebuffer(), pt.accumulator())
    pt.render(spp=2, bounces=4, seed=3, first_sample=1, reset=True)
    pt.render(spp=1, bounces=4, seed=3, first_sample=3, reset=False)
    pt.render(spp=4, bounces=4, seed=3, first_sample=4, reset=False)
    assert np.array_equal(pt.framebuffer(), one[0])
    assert np.array_equal(pt.accumulator().view(np.uint32), one[1].view(np.uint32))
    ofb, oacc, _ = _oracle_frame(oracle, objs, n, w, h, spp=7, bounces=4, seed=3)
    assert np.array_equal(one[0], ofb) and np.array_equal(one[1].view(np.uint32), oacc.view(np.uint32))


def test_write_accumulator_then_resume(srt, oracle):
    """Accumulator upload (incl. a non-zero alpha) continues exactly like the oracle."""
    w, h = 48, 40
    pt, objs, n = _pt(srt, "Scene1", w, h)
    rng = np.random.default_rng(1)
    acc0 = rng.random((h, w, 4), dtype=np.float32) * 3
    pt.write_accumulator(acc0)
    pt.render(spp=2, bounces=3, seed=9, first_sample=5, reset=False)
    ofb, oacc, _ = _oracle_frame(oracle, objs, n, w, h, spp=2, bounces=3, seed=9, first_sample=5, reset=False, accumulator=acc0)
    assert np.array_equal(pt.accumulator().view(np.uint32), oacc.view(np.uint32))
    assert np.array_equal(pt.framebuffer(), ofb)
    assert ((pt.framebuffer() >> 24) == 255).all()  # a/(0+a) = 1 -> alpha byte 255


def test_bind_output_renders_into_caller_memory(srt):
    import torch

    w, h = 64, 32
    pt, objs, n = _pt(srt, "Scene3", w, h)
    pt.render(spp=2, bounces=2, seed=0)
    own = pt.framebuffer()
    t = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    pt.bind_output(d_framebuffer=t.data_ptr())
    pt.render(spp=2, bounces=2, seed=0)
    pt.wait()
    assert np.array_equal(t.cpu().numpy().view(np.uint32), own)
    pt.bind_output()


def test_edge_cases_empty_scene_and_errors(srt, oracle):
    w, h = 33, 17
    pt = srt.PathTracer(w, h)
    with pytest.raises(srt.SrtError) as e:
        pt.render()
    assert e.value.code == srt.capi.ERR_STATE
    pt.set_scene((srt.Object * 1)(), 0)  # empty ObjectsToRender: every pixel is sky
    pt.set_camera(srt.default_camera())
    pt.render(spp=2, bounces=8, seed=0, count_rays=True)
    arr, n = oracle.make_objects([])
    ofb, oacc, orays = oracle.render(arr, 0, oracle.default_environment(), oracle.default_camera(), w, h, spp=2, bounces=8, seed=0)
    assert np.array_equal(pt.framebuffer(), ofb) and pt.stats().rays == orays == w * h * 2
    for bad in [dict(rows=(5, 5)), dict(rows=(0, h + 1)), dict(spp=0), dict(first_sample=0), dict(bounces=-1)]:
        with pytest.raises(srt.SrtError):
            pt.render(**bad)
    assert pt.poll() in (True, False)
    pt.wait()
    assert pt.poll() is True
    # inert objects keep list slots but never hit; a box before a coincident sphere wins the tie
    objs = [dict(type=oracle.OBJ_NONE, position=(0, 0, 3)),
            dict(type=oracle.OBJ_SPHERE, position=(0.3, 0.2, 6), radius=1.5, emissive=(1, 0, 0)),
            dict(type=oracle.OBJ_BOX, position=(0.3, 0.2, 6), half_size=(1, 1, 1), emissive=(0, 1, 0)),
            dict(type=oracle.OBJ_BOX, position=(0.3, 0.2, 6), half_size=(1, 1, 1), emissive=(0, 0, 1))]
    arr, n = oracle.make_objects(objs)
    pt.set_scene(C.cast(arr, C.POINTER(srt.Object)), n)
    pt.render(spp=2, bounces=3, seed=1)
    ofb, oacc, _ = oracle.render(arr, n, oracle.default_environment(), oracle.default_camera(), w, h, spp=2, bounces=3, seed=1)
    assert np.array_equal(pt.framebuffer(), ofb) and np.array_equal(pt.accumulator().view(np.uint32), oacc.view(np.uint32))


def test_camera_moved_and_rotated(srt, oracle):
    w, h = 120, 80
    pt, objs, n = _pt(srt, "Scene_indirect", w, h)
    basis = srt.host.rotate_about_axis([1, 0, 0, 0, 1, 0, 0, 0, 1], 0.4, (0, 1, 0))
    basis = srt.host.rotate_about_axis(basis, -0.15, basis[0:3])
    cam = srt.Camera()
    cam.position = (C.c_float * 3)(0.5, 0.3, -1.0)
    cam.right, cam.up, cam.forward = (C.c_float * 3)(*basis[0:3]), (C.c_float * 3)(*basis[3:6]), (C.c_float * 3)(*basis[6:9])
    cam.fov_degrees = 90
    pt.set_camera(cam)
    pt.render(spp=2, bounces=6, seed=2)
    ocam = oracle.Camera.from_buffer_copy(bytes(cam))
    ofb, oacc, _ = oracle.render(C.cast(objs, C.POINTER(oracle.Object)), n, oracle.default_environment(), ocam, w, h, spp=2, bounces=6, seed=2)
    assert np.array_equal(pt.framebuffer(), ofb) and np.array_equal(pt.accumulator().view(np.uint32), oacc.view(np.uint32))


def test_custom_environment(srt, oracle):
    w, h = 64, 64
    pt, objs, n = _pt(srt, "Scene1", w, h)
    env = srt.default_environment()
    env.sky_color = (C.c_float * 3)(0.1, 0.2, 0.9)
    env.sun_direction = (C.c_float * 3)(0.0, -0.6, 0.8)
    env.sun_color = (C.c_float * 3)(30, 20, 10)
    pt.set_environment(env)
    pt.render(spp=2, bounces=4, seed=0)
    oenv = oracle.Environment.from_buffer_copy(bytes(env))
    ofb, oacc, _ = oracle.render(C.cast(objs, C.POINTER(oracle.Object)), n, oenv, oracle.default_camera(), w, h, spp=2, bounces=4, seed=0)
    assert np.array_equal(pt.framebuffer(), ofb) and np.array_equal(pt.accumulator().view(np.uint32), oacc.view(np.uint32))
    # The environment lives in the scene image's constants block (round 3): set BEFORE the scene it must survive srt_set_scene,
    # changed again AFTER it must be patched in place; negative components go through Color's clamp (Common.hpp:253-262);
    # the preview shader reads the same rows.
    pt2 = srt.PathTracer(w, h)
    env.ground_color = (C.c_float * 3)(-0.5, 0.3, 0.0)
    env.horizon_color = (C.c_float * 3)(2.0, -1.0, 0.25)
    pt2.set_environment(env)
    pt2.set_scene(objs, n)
    pt2.set_camera(srt.default_camera())
    oenv = oracle.Environment.from_buffer_copy(bytes(env))
    for kw in (dict(spp=3, bounces=5), dict(spp=1, bounces=2, preview=True)):
        pt2.render(seed=7, **kw)
        ofb, oacc, _ = oracle.render(C.cast(objs, C.POINTER(oracle.Object)), n, oenv, oracle.default_camera(), w, h, seed=7, **kw)
        assert np.array_equal(pt2.framebuffer(), ofb) and np.array_equal(pt2.accumulator().view(np.uint32), oacc.view(np.uint32))
    env.sun_color = (C.c_float * 3)(-5, 400, 1)
    env.sky_color = (C.c_float * 3)(3, 3, 3)
    pt2.set_environment(env)
    pt2.render(spp=2, bounces=3, seed=1)
    oenv = oracle.Environment.from_buffer_copy(bytes(env))
    ofb, oacc, _ = oracle.render(C.cast(objs, C.POINTER(oracle.Object)), n, oenv, oracle.default_camera(), w, h, spp=2, bounces=3, seed=1)
    assert np.array_equal(pt2.framebuffer(), ofb) and np.array_equal(pt2.accumulator().view(np.uint32), oacc.view(np.uint32))
    pt2.close()


def test_host_renderer_sequences(srt, oracle):
    """C++ PathTraceRenderer: the clean sequence and the reference's UI sequence
    (Raytracer.cpp:572-590: first full frame after an edit has setFrame AND ACC == 2)."""
    w, h = 80, 60
    scene = srt.host.Scene(scene_path("Scene3_indirect"))
    objs, n = scene.objects_copy()
    r = srt.host.Renderer(w, h)
    r.settings(fov=55, max_bounces=3, target_frames=6, seed=4)
    r.set_scene(scene)
    r.render_samples(2)
    r.render_samples(3, count_rays=True)
    ofb, oacc, _ = _oracle_frame(oracle, objs, n, w, h, spp=5, bounces=3, seed=4)
    assert np.array_equal(r.framebuffer(), ofb) and np.array_equal(r.accumulator().view(np.uint32), oacc.view(np.uint32))
    assert r.accumulation_frames == 5 and r.stats().path_samples == w * h * 3
    # UI sequence in path-trace mode at full scale (Raytracer.cpp:572-590): after an edit the loop
    # renders f=1 at 1/4 resolution (steps 4, reset), then f=2 full resolution WITH reset (the
    # quirk), then f=3..6 accumulating, then stops at TARGETFRAMES.
    r.mode(simpledraw=False, screen_scale=1.0)
    launched = 0
    while r.render_frame():
        launched += 1
    assert launched == 6 and r.accumulation_frames == 6 and not r.render_frame()
    _, oacc, _ = _oracle_frame(oracle, objs, n, w, h, spp=1, bounces=3, seed=4, first_sample=2, reset=True)
    ofb, oacc, _ = _oracle_frame(oracle, objs, n, w, h, spp=4, bounces=3, seed=4, first_sample=3, reset=False, accumulator=oacc)
    assert np.array_equal(r.framebuffer(), ofb) and np.array_equal(r.accumulator().view(np.uint32), oacc.view(np.uint32))
    # row band through the host class
    r.set_band(10, 30)
    r.render_samples(2)
    ofb2, _, _ = _oracle_frame(oracle, objs, n, w, h, spp=2, bounces=3, seed=4)
    assert np.array_equal(r.framebuffer(), ofb2[10:30])
    r.close()


def test_full_size_properties_1080p(srt, oracle):
    """BASELINE configs[1] size. Oracle compare at 1 spp (seconds on the host cores), then
    size-independent properties at the full 32 spp: determinism across launches, band
    concatenation, resume == one-shot."""
    w, h = 1920, 1080
    pt, objs, n = _pt(srt, "Scene1", w, h)
    pt.render(spp=1, bounces=8, seed=0, count_rays=True)
    fb1 = pt.framebuffer()
    ofb, oacc, orays = _oracle_frame(oracle, objs, n, w, h, spp=1, bounces=8, seed=0, threads=16)
    assert np.array_equal(fb1, ofb) and pt.stats().rays == orays
    assert np.array_equal(pt.accumulator().view(np.uint32), oacc.view(np.uint32))
    pt.render(spp=32, bounces=8, seed=0)
    a = pt.framebuffer()
    pt.render(spp=32, bounces=8, seed=0)
    assert np.array_equal(pt.framebuffer(), a)  # deterministic
    for rb, re in [(0, 135), (135, 600), (600, 1080)]:
        pt.render(spp=32, bounces=8, seed=0, rows=(rb, re))
        assert np.array_equal(pt.framebuffer(rows=(rb, re)), a[rb:re])
    pt.render(spp=20, bounces=8, seed=0)
    pt.render(spp=12, bounces=8, seed=0, first_sample=21, reset=False)
    assert np.array_equal(pt.framebuffer(), a)
    assert (a >> 24 == 0).all()  # alpha byte is always 0 (Raytracer.cpp:74)


@pytest.mark.parametrize("name", ["Scene1", "Scene3", "Scene_indirect"])
def test_preview_steps_and_picking(srt, oracle, name):
    """SIMPLEDRAW shader (:147-160), progressive blocks (:233-248), picking (:525-541)."""
    w, h = 150, 84
    pt, objs, n = _pt(srt, name, w, h)
    oarr = C.cast(objs, C.POINTER(oracle.Object))
    env, cam = oracle.default_environment(), oracle.default_camera()
    div = w // 16 + 1
    cases = [dict(preview=True), dict(preview=True, selected=5), dict(preview=True, steps=2, stripe_width=div),
             dict(preview=False, steps=4, stripe_width=div), dict(preview=False, steps=8, stripe_width=0),
             dict(preview=True, steps=3, stripe_width=div, selected=n - 1)]
    for kw in cases:
        pt.render(spp=2, bounces=4, seed=3, **kw)
        ofb, oacc, _ = oracle.render(oarr, n, env, cam, w, h, spp=2, bounces=4, seed=3, **kw)
        assert np.array_equal(pt.framebuffer(), ofb), kw
        assert np.array_equal(pt.accumulator().view(np.uint32), oacc.view(np.uint32)), kw
    # picking equals the oracle's GetClosestObject on the same pixel ray
    import ctypes
    d, nn, pp, t = (ctypes.c_float * 3)(), (ctypes.c_float * 3)(), (ctypes.c_float * 3)(), ctypes.c_float()
    origin = (ctypes.c_float * 3)(0, 0, 0)
    hits = 0
    for (x, y) in [(0, 0), (75, 42), (20, 10), (140, 70), (75, 5), (33, 33), (100, 20), (149, 83)]:
        oracle.lib().srt_oracle_ray_direction(ctypes.byref(cam), w, h, x, y, d)
        idx = oracle.lib().srt_oracle_closest(oarr, n, origin, d, nn, pp, ctypes.byref(t))
        assert pt.pick(x, y) == idx
        hits += idx >= 0
    assert hits >= 2
    pt.close()


def test_host_renderer_default_ui_loop(srt, oracle):
    """A fresh PathTraceRenderer behaves like the reference at start-up: SIMPLEDRAW preview at
    SCREEN_SCALE .5 -> steps 2 blocks anchored at 16 stripes; an edit triggers a steps-8 frame."""
    w, h = 128, 72
    scene = srt.host.Scene(scene_path("Scene1"))
    objs, n = scene.objects_copy()
    oarr = C.cast(objs, C.POINTER(oracle.Object))
    env, cam = oracle.default_environment(), oracle.default_camera()
    r = srt.host.Renderer(w, h)
    r.set_scene(scene)
    div = w // 16 + 1
    assert r.render_frame()  # start-up globals: ACC 1, no reset, steps = ceil(1/.5) = 2
    ofb, oacc, _ = oracle.render(oarr, n, env, cam, w, h, spp=1, bounces=2, seed=0, preview=True, steps=2, stripe_width=div, reset=False)
    assert np.array_equal(r.framebuffer(), ofb) and np.array_equal(r.accumulator().view(np.uint32), oacc.view(np.uint32))
    assert r.render_frame()  # SetScene raised doSetFrame: setFrame, scaler 1/4 -> steps = ceil(1/(.5*.25)) = 8
    ofb, oacc, _ = oracle.render(oarr, n, env, cam, w, h, spp=1, bounces=2, seed=0, preview=True, steps=8, stripe_width=div, reset=True)
    assert np.array_equal(r.framebuffer(), ofb)
    assert r.render_frame() and r.accumulation_frames == 1  # preview never advances ACCUMULATIONFRAMES (:590)
    # picking through the host class uses window coordinates (y down)
    idx = r.pick(64, 36)
    assert idx == 64  # the big ball at (0,0,5) is object 64 of Scene1 and fills the centre
    r.close()


@pytest.mark.parametrize("name,w,h,spp,tile", [("Scene1", 1920, 300, 16, 4), ("Scene_indirect", 1920, 128, 16, 2), ("Scene3", 208, 77, 24, 1),
                                               ("Scene1_reflection", 640, 50, 40, 1), ("Scene1", 320, 64, 64, 0), ("Scene_indirect", 200, 41, 80, 0),
                                               ("Scene2", 1000, 30, 100, 0)])
def test_small_tiles_and_sample_chunks_at_high_sample_counts(srt, oracle, name, w, h, spp, tile):
    """With >= 16 samples per pixel and too few rows to fill the chip, srt_render shrinks a wave's pixel
    tile (8 rows -> 4 / 2 / 1) and uses the kernel instantiation whose path pool hands several samples
    of one pixel out at once; from 64 samples per pixel it splits the samples of a tile over several
    workgroups and folds the stored colours in a second kernel (narrow stripes of a multi-GPU frame).
    `tile` is what the rule picks for these sizes on 256 CUs (0 = sample chunks); whatever it picks, the
    bits must equal the oracle's — also when resuming on a ragged band."""
    pt, objs, n = _pt(srt, name, w, h)
    pt.render(spp=spp, bounces=6, seed=2, count_rays=True)
    ofb, oacc, orays = _oracle_frame(oracle, objs, n, w, h, spp=spp, bounces=6, seed=2)
    assert pt.stats().rays == orays
    assert np.array_equal(pt.accumulator().view(np.uint32), oacc.view(np.uint32))
    assert np.array_equal(pt.framebuffer(), ofb)
    # resume with another high-count launch on a ragged band
    more = 17 if spp < 64 else 70
    pt.render(spp=more, bounces=6, seed=2, first_sample=spp + 1, reset=False, rows=(3, h - 5))
    ofb2, oacc2, _ = _oracle_frame(oracle, objs, n, w, h, spp=more, bounces=6, seed=2, first_sample=spp + 1, reset=False, accumulator=oacc,
                                   rows=(3, h - 5))
    assert np.array_equal(pt.accumulator().view(np.uint32), oacc2.view(np.uint32))
    assert np.array_equal(pt.framebuffer(rows=(3, h - 5)), ofb2[3:h - 5])
    pt.close()


def test_progressive_blocks_with_sample_chunks(srt, oracle):
    """steps x steps blocks (Raytracer.cpp:233-248) together with a sample-chunked launch (>= 64 spp)."""
    w, h = 150, 84
    pt, objs, n = _pt(srt, "Scene3", w, h)
    kw = dict(spp=72, bounces=4, seed=9, steps=2, stripe_width=w // 16 + 1)
    pt.render(**kw)
    ofb, oacc, _ = _oracle_frame(oracle, objs, n, w, h, **kw)
    assert np.array_equal(pt.framebuffer(), ofb)
    assert np.array_equal(pt.accumulator().view(np.uint32), oacc.view(np.uint32))
    pt.close()


@pytest.mark.parametrize("name", ["Scene3", "Scene_indirect"])
def test_block_grid_launches(srt, oracle, name):
    """Progressive blocks (Raytracer.cpp:233-248) through the launches whose lanes are BLOCKS (srt_render picks them when the
    launch starts the frame or adds one sample): odd frame sizes, stripes that the block size does not divide, blocks of more
    than 64 pixels, row bands that cut blocks, one sample folded onto an accumulated frame — and the one-lane-per-pixel
    fallback (several samples onto an accumulated frame).  Bits against the oracle; one ray per block and bounce."""
    w, h = 157, 91
    pt, objs, n = _pt(srt, name, w, h)
    for steps in (2, 3, 5, 8, 11):
        for sw in (0, w // 16 + 1):
            kw = dict(bounces=4, seed=17, steps=steps, stripe_width=sw)
            # the launch starts the frame: one and several samples
            for spp in (1, 3):
                pt.render(spp=spp, count_rays=True, **kw)
                ofb, oacc, orays = _oracle_frame(oracle, objs, n, w, h, spp=spp, **kw)
                assert np.array_equal(pt.framebuffer(), ofb), (steps, sw, spp)
                assert np.array_equal(pt.accumulator().view(np.uint32), oacc.view(np.uint32)), (steps, sw, spp)
                # (the oracle's default walk evaluates the block's ray for every pixel; renderArea's own loop nest — the
                # literal mode, 16 column stripes, one sample — traces once per block, and so does the block grid)
                if spp == 1 and sw > 0:
                    lfb, _, lrays = _oracle_frame(oracle, objs, n, w, h, spp=1, split=oracle.SPLIT_REF_COLS, threads=16, **kw)
                    assert np.array_equal(lfb, ofb) and pt.stats().rays == lrays, (steps, sw, pt.stats().rays, lrays, orays)
            # a full-resolution frame of two samples, then ONE sample in blocks folded onto it (every pixel its own mean)
            pt.render(spp=2, bounces=4, seed=17)
            _, acc2, _ = _oracle_frame(oracle, objs, n, w, h, spp=2, bounces=4, seed=17)
            pt.render(spp=1, first_sample=3, reset=False, **kw)
            ofb, oacc, _ = _oracle_frame(oracle, objs, n, w, h, spp=1, first_sample=3, reset=False, accumulator=acc2, **kw)
            assert np.array_equal(pt.framebuffer(), ofb), (steps, sw, "keep")
            assert np.array_equal(pt.accumulator().view(np.uint32), oacc.view(np.uint32)), (steps, sw, "keep")
            # several samples onto that frame: one lane per pixel
            pt.render(spp=3, first_sample=4, reset=False, **kw)
            ofb, oacc2, _ = _oracle_frame(oracle, objs, n, w, h, spp=3, first_sample=4, reset=False, accumulator=oacc, **kw)
            assert np.array_equal(pt.framebuffer(), ofb), (steps, sw, "fallback")
            assert np.array_equal(pt.accumulator().view(np.uint32), oacc2.view(np.uint32)), (steps, sw, "fallback")
    # row bands that cut through blocks, rendered one after the other, give the single-launch frame
    kw = dict(spp=1, bounces=4, seed=5, steps=5, stripe_width=w // 16 + 1)
    pt.render(**kw)
    full_fb, full_acc = pt.framebuffer(), pt.accumulator()
    pt2 = srt.PathTracer(w, h)
    pt2.set_scene(objs, n)
    pt2.set_camera(srt.default_camera())
    for rb, re in [(0, 1), (1, 16), (16, 17), (17, 64), (64, 91)]:
        pt2.render(rows=(rb, re), **kw)
    assert np.array_equal(pt2.framebuffer(), full_fb)
    assert np.array_equal(pt2.accumulator().view(np.uint32), full_acc.view(np.uint32))
    pt2.close()
    pt.close()


@pytest.mark.parametrize("name,w,h,spp", [("Scene1", 800, 450, 6), ("Scene3", 1024, 400, 5)])
def test_mid_size_frames_in_cost_order(srt, oracle, name, w, h, spp):
    """Frames of 512..2048 blocks of tiles are dispatched in cost order too (first launch: the device-side estimate, later
    launches: recorded costs): every launch gives the oracle's bits and ray count."""
    pt, objs, n = _pt(srt, name, w, h)
    ofb, oacc, orays = _oracle_frame(oracle, objs, n, w, h, spp=spp, bounces=8, seed=11)
    for launch in range(4):
        pt.render(spp=spp, bounces=8, seed=11, count_rays=True)
        assert np.array_equal(pt.framebuffer(), ofb), launch
        assert np.array_equal(pt.accumulator().view(np.uint32), oacc.view(np.uint32)), launch
        assert pt.stats().rays == orays, launch
    pt.close()


def test_sample_chunks_extremes(srt, oracle):
    """Sample-chunked launches at the edges: thousands of samples on a frame smaller than one workgroup
    (hundreds of chunks per tile), and a band without a single traced pixel (all sky: every tile mask 0)."""
    w, h = 13, 7
    pt, objs, n = _pt(srt, "Scene1", w, h)
    kw = dict(spp=4100, bounces=3, seed=21)
    pt.render(count_rays=True, **kw)
    ofb, oacc, orays = _oracle_frame(oracle, objs, n, w, h, **kw)
    assert pt.stats().rays == orays
    assert np.array_equal(pt.accumulator().view(np.uint32), oacc.view(np.uint32))
    assert np.array_equal(pt.framebuffer(), ofb)
    pt.close()
    # a frame without a single traced pixel: every tile mask is 0, the fold kernel has nothing to do
    w, h = 320, 180
    oarr, n = oracle.make_objects([dict(type=oracle.OBJ_SPHERE, position=(0.0, 0.0, -50.0), radius=1.0)])  # behind the camera
    pt = srt.PathTracer(w, h)
    pt.set_scene(C.cast(oarr, C.POINTER(srt.Object)), n)
    pt.set_camera(srt.default_camera())
    kw = dict(spp=96, bounces=8, seed=1, rows=(10, 150))
    pt.render(count_rays=True, **kw)
    ofb, oacc, orays = oracle.render(oarr, n, oracle.default_environment(), oracle.default_camera(), w, h, **kw)
    assert pt.stats().rays == orays == w * 140 * 96  # one (missing) primary ray per sample
    assert np.array_equal(pt.accumulator().view(np.uint32), oacc.view(np.uint32))
    assert np.array_equal(pt.framebuffer(rows=(10, 150)), ofb[10:150])
    pt.close()


@pytest.mark.parametrize("name,w,h,spp,more", [("Scene1", 640, 64, 65, 97), ("Scene_indirect", 320, 48, 100, 131), ("Scene3", 1920, 32, 257, 64)])
def test_tapered_chunks_with_awkward_sample_counts(srt, oracle, name, w, h, spp, more):
    """Sample-chunked launches end in half-size chunks (the launch's last workgroups, KernelParams.chunk_full): sample counts
    that leave the full chunks, the half chunks and the last chunk all different (65 = 33 + 16 + 16, 257, a resumed frame whose
    first_sample is not 1); at least two launches each, so that the second runs with the recorded costs' chunk count."""
    pt, objs, n = _pt(srt, name, w, h)
    for launch in range(2):
        pt.render(spp=spp, bounces=6, seed=5, count_rays=True)
        assert pt.stats().sample_chunks >= 2
    ofb, oacc, orays = _oracle_frame(oracle, objs, n, w, h, spp=spp, bounces=6, seed=5)
    assert pt.stats().rays == orays
    assert np.array_equal(pt.accumulator().view(np.uint32), oacc.view(np.uint32)) and np.array_equal(pt.framebuffer(), ofb)
    pt.render(spp=more, bounces=6, seed=5, first_sample=spp + 1, reset=False)
    assert pt.stats().sample_chunks >= 2
    ofb2, oacc2, _ = _oracle_frame(oracle, objs, n, w, h, spp=more, bounces=6, seed=5, first_sample=spp + 1, reset=False, accumulator=oacc)
    assert np.array_equal(pt.accumulator().view(np.uint32), oacc2.view(np.uint32)) and np.array_equal(pt.framebuffer(), ofb2)
    pt.close()


def test_poll_while_busy_then_render_again(srt):
    """srt_poll on a running launch reports "not done" without poisoning the next call: HIP's not-ready
    status must not come back from hipGetLastError() as that launch's error (same for the internal poll
    of the block-cost feedback)."""
    pt, objs, n = _pt(srt, "Scene_indirect", 1920, 1080)
    pt.render(spp=32, bounces=8, seed=0)   # ~20 ms on an MI355X
    first = pt.poll()
    pt.render(spp=1, bounces=8, seed=0)    # enqueued behind it; must not fail
    second = pt.poll()
    pt.render(spp=1, bounces=8, seed=0)
    pt.wait()
    assert pt.poll() is True
    assert first is False or second is False or True  # timing-dependent; the calls above not raising is the test
    a = pt.framebuffer()
    pt.render(spp=1, bounces=8, seed=0)
    assert np.array_equal(pt.framebuffer(), a)
    pt.close()


@pytest.mark.parametrize("name,rows,spp,mesh", [("Scene1", (945, 1080), 256, 0), ("Scene1", (540, 675), 384, 0), ("Scene3", (400, 540), 256, 0)])
def test_a_sample_chunked_launch_equals_the_same_samples_in_small_launches(srt, name, rows, spp, mesh):
    """Chained chunks: in a sample-chunked launch a chunk folds its own samples when it finds its tile's running
    mean at its own first sample, and stores them for fold_kernel otherwise — which of the two depends on when the workgroups
    happen to run (on a grid that barely fills the chip about half the tiles chain).  Whatever they did, the running mean is
    folded in sample order: the band must come out bit for bit as from a sequence of 32-sample launches, each of which resumes
    the frame in one piece — three times over, in fresh contexts."""
    w, h = 1920, 1080
    sc = srt.host.Scene(scene_path(name))
    objs, n = sc.objects_copy()

    def band(chunks_of):
        pt = srt.PathTracer(w, h)
        pt.set_scene(objs, n)
        pt.set_camera(srt.default_camera())
        done, layers = 0, []
        while done < spp:
            k = min(chunks_of, spp - done)
            pt.render(spp=k, bounces=8, seed=21, first_sample=1 + done, reset=done == 0, rows=rows)
            layers.append(int(pt.stats().sample_chunks))
            done += k
        fb, acc = pt.framebuffer().copy(), pt.accumulator().copy()  # (whole frames: outside the band both are untouched)
        pt.close()
        return fb, acc, layers

    ref_fb, ref_acc, small = band(32)
    assert max(small) <= 2  # (32 samples: one piece, or the two halves of an analytic launch)
    for _ in range(3):
        fb, acc, layers = band(spp)
        assert layers[0] >= 4, layers  # a sample-chunked launch
        assert np.array_equal(acc.view(np.uint32), ref_acc.view(np.uint32))
        assert np.array_equal(fb, ref_fb)
