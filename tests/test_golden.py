"""Committed golden fixtures (tests/golden/frames.json, made by make_golden.py from the
oracle): the oracle must keep reproducing them on the CPU; the HIP path must reproduce them
on the GPU with no oracle in the loop."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN, scene_path

G = json.load(open(os.path.join(GOLDEN, "frames.json")))
FRAMES = G["frames"]
IDS = ["%s-%dx%d-s%d-b%d" % (f["scene"], f["width"], f["height"], f["spp"], f["bounces"]) for f in FRAMES]


def _crop(f, fb):
    c = G["crop"]
    y0, x0 = (f["height"] - c) // 2, (f["width"] - c) // 2
    return fb[y0:y0 + c, x0:x0 + c]


@pytest.mark.parametrize("f", FRAMES, ids=IDS)
def test_oracle_reproduces_golden(oracle, f):
    arr, n = oracle.make_objects(oracle.load_scene_json_py(scene_path(f["scene"])))
    fb, acc, rays = oracle.render(arr, n, oracle.default_environment(), oracle.default_camera(), f["width"], f["height"],
                                  spp=f["spp"], bounces=f["bounces"], seed=f["seed"], pow_mode=oracle.POW_SHARED)
    assert rays == f["rays"]
    assert _crop(f, fb).astype("<u4").tobytes().hex() == f["fb_crop_hex"]
    assert oracle.frame_hash(fb) == f["fb_sha"] and oracle.frame_hash(acc) == f["acc_sha"]


def test_oracle_thread_split_invariance(oracle):
    """Row bands, reference column stripes and 1 thread give the same frame (pixels are independent)."""
    f = FRAMES[1]
    arr, n = oracle.make_objects(oracle.load_scene_json_py(scene_path(f["scene"])))
    kw = dict(spp=f["spp"], bounces=f["bounces"], seed=f["seed"])
    env, cam = oracle.default_environment(), oracle.default_camera()
    a = oracle.render(arr, n, env, cam, f["width"], f["height"], threads=1, **kw)
    b = oracle.render(arr, n, env, cam, f["width"], f["height"], threads=16, split=oracle.SPLIT_REF_COLS, **kw)
    c = oracle.render(arr, n, env, cam, f["width"], f["height"], threads=7, **kw)
    for x in (b, c):
        assert np.array_equal(a[0], x[0]) and np.array_equal(a[1], x[1]) and a[2] == x[2]
    assert oracle.frame_hash(a[0]) == f["fb_sha"]


def test_oracle_progressive_equals_one_shot(oracle):
    """spp in one call == the same samples over several resumed calls (clean sequence)."""
    f = FRAMES[1]
    arr, n = oracle.make_objects(oracle.load_scene_json_py(scene_path(f["scene"])))
    env, cam = oracle.default_environment(), oracle.default_camera()
    w, h = f["width"], f["height"]
    fb1, acc1, _ = oracle.render(arr, n, env, cam, w, h, spp=1, bounces=f["bounces"], seed=f["seed"], first_sample=1, reset=True)
    fb2, acc2, _ = oracle.render(arr, n, env, cam, w, h, spp=3, bounces=f["bounces"], seed=f["seed"], first_sample=2, reset=False, accumulator=acc1)
    assert oracle.frame_hash(fb2) == f["fb_sha"] and oracle.frame_hash(acc2) == f["acc_sha"]


def test_ray_direction_bits_pinned(oracle):
    cam = oracle.default_camera()
    d = (C.c_float * 3)()
    oracle.lib().srt_oracle_ray_direction(C.byref(cam), 1920, 1080, 0, 0, d)
    assert [struct.unpack("<I", struct.pack("<f", v))[0] for v in d] == G["ray_dir_1080p_pixel00_bits"]


@pytest.mark.gpu
@pytest.mark.parametrize("f", FRAMES, ids=IDS)
def test_hip_reproduces_golden(srt, f):
    import hashlib

    s = srt.host.Scene(scene_path(f["scene"]))
    objs, n = s.objects_copy()
    with srt.PathTracer(f["width"], f["height"]) as pt:
        pt.set_scene(objs, n)
        pt.set_camera(srt.default_camera())
        pt.render(spp=f["spp"], bounces=f["bounces"], seed=f["seed"], count_rays=True)
        fb, acc, st = pt.framebuffer(), pt.accumulator(), pt.stats()
    assert st.rays == f["rays"]
    assert _crop(f, fb).astype("<u4").tobytes().hex() == f["fb_crop_hex"]
    assert hashlib.sha256(fb.tobytes()).hexdigest()[:16] == f["fb_sha"]
    assert hashlib.sha256(acc.tobytes()).hexdigest()[:16] == f["acc_sha"]
