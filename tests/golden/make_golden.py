#!/usr/bin/env python3
"""Generates tests/golden/frames.json from the CPU oracle.

What these fixtures are: REGRESSION pins of this project's oracle (and therefore of the
RNG stream, srt_powf and the restated arithmetic) — hashes of full frames plus small
crops, so that a drift of the oracle across compilers / machines / edits is caught on the
CPU, and the HIP path can be checked on the GPU box without the oracle in the loop.
What they are NOT: outputs of the reference.  The reference has no golden data and cannot
be built in this image, so parity with it remains unpinned.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import srt_oracle_py as O  # noqa: E402

SCENES = ["Scene1", "Scene1_reflection", "Scene2", "Scene3", "Scene3_indirect", "Scene_indirect"]
CONFIGS = [  # (W, H, spp, bounces, seed)
    (256, 256, 1, 4, 0),   # BASELINE.json configs[0]
    (160, 90, 4, 8, 0),
    (128, 72, 16, 2, 7),
]
CROP = 24


def main():
    env, cam = O.default_environment(), O.default_camera()
    out = {"note": "oracle-generated regression fixtures (POW_SHARED); not reference output", "crop": CROP, "frames": []}
    for name in SCENES:
        arr, n = O.make_objects(O.load_scene_json_py(os.path.join(ROOT, "software-raytracer_amd", "scenes", name + ".json")))
        for (w, h, spp, b, seed) in CONFIGS:
            fb, acc, rays = O.render(arr, n, env, cam, w, h, spp=spp, bounces=b, seed=seed, pow_mode=O.POW_SHARED)
            lfb, _, _ = O.render(arr, n, env, cam, w, h, spp=spp, bounces=b, seed=seed, pow_mode=O.POW_LIBM)
            y0, x0 = (h - CROP) // 2, (w - CROP) // 2
            crop_fb = fb[y0:y0 + CROP, x0:x0 + CROP]
            crop_acc = acc[h - 1 - (y0 + CROP - 1):h - y0, x0:x0 + CROP]  # same pixels, scene rows
            out["frames"].append({
                "scene": name, "width": w, "height": h, "spp": spp, "bounces": b, "seed": seed,
                "rays": rays,
                "fb_sha": O.frame_hash(fb), "acc_sha": O.frame_hash(acc),
                "fb_crop_hex": crop_fb.astype("<u4").tobytes().hex(),
                "acc_crop_sha": O.frame_hash(np.ascontiguousarray(crop_acc)),
                "libm_pow_pixels_differing": int((fb != lfb).sum()),
            })
            print(name, w, h, spp, b, out["frames"][-1]["fb_sha"], rays, out["frames"][-1]["libm_pow_pixels_differing"])
    # frame-constant camera scalars of GetRayDirection for FOV 55 (Raytracer.cpp:112-115), as float bits
    import ctypes as C
    d = (C.c_float * 3)()
    O.lib().srt_oracle_ray_direction(C.byref(cam), 1920, 1080, 0, 0, d)
    out["ray_dir_1080p_pixel00_bits"] = [struct.unpack("<I", struct.pack("<f", v))[0] for v in d]
    with open(os.path.join(HERE, "frames.json"), "w") as f:
        json.dump(out, f, indent=0)


if __name__ == "__main__":
    main()
