"""ctypes wrapper of the CPU oracle (oracle/libsrt_oracle.so).  TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg —
never by the product package.  PARITY UNPINNED (see srt_oracle.h / DESIGN.md §3).

Also holds an independent, deliberately naive Python reader for the reference's scene
JSON (Raytracer/Scene.hpp:27-80) so that tests can cross-check the product's C++
loader against a second implementation.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsrt_oracle.so")

POW_LIBM = 0
POW_SHARED = 1
SPLIT_ROWS = 0
SPLIT_REF_COLS = 1
RENDER_RESET = 1
RENDER_COUNT_RAYS = 2
RENDER_PREVIEW = 4

OBJ_NONE, OBJ_SPHERE, OBJ_BOX, OBJ_MESH = 0, 1, 2, 3


class Material(C.Structure):
    _fields_ = [
        ("smoothness", C.c_float),
        ("specular_amount", C.c_float),
        ("base_color", C.c_float * 3),
        ("emissive_color", C.c_float * 3),
        ("specular_color", C.c_float * 3),
    ]


class Object(C.Structure):
    _fields_ = [
        ("type", C.c_int32),
        ("position", C.c_float * 3),
        ("radius", C.c_float),
        ("half_size", C.c_float * 3),
        ("material", Material),
        ("mesh", C.c_int32),
    ]


class Mesh(C.Structure):
    _fields_ = [
        ("vertices", C.POINTER(C.c_float)),
        ("vertex_count", C.c_size_t),
        ("indices", C.POINTER(C.c_uint32)),
        ("triangle_count", C.c_size_t),
    ]


class Environment(C.Structure):
    _fields_ = [
        ("sun_direction", C.c_float * 3),
        ("sky_color", C.c_float * 3),
        ("horizon_color", C.c_float * 3),
        ("ground_color", C.c_float * 3),
        ("sun_color", C.c_float * 3),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("position", C.c_float * 3),
        ("right", C.c_float * 3),
        ("up", C.c_float * 3),
        ("forward", C.c_float * 3),
        ("fov_degrees", C.c_int32),
    ]


class RenderParams(C.Structure):
    _fields_ = [
        ("row_begin", C.c_int32),
        ("row_end", C.c_int32),
        ("first_sample", C.c_uint32),
        ("sample_count", C.c_uint32),
        ("max_bounces", C.c_int32),
        ("seed", C.c_uint32),
        ("flags", C.c_uint32),
        ("steps", C.c_int32),
        ("stripe_width", C.c_int32),
        ("selected_object", C.c_int32),
    ]


class Job(C.Structure):
    _fields_ = [
        ("objects", C.POINTER(Object)),
        ("object_count", C.c_size_t),
        ("meshes", C.POINTER(Mesh)),
        ("mesh_count", C.c_size_t),
        ("env", C.POINTER(Environment)),
        ("camera", C.POINTER(Camera)),
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("params", RenderParams),
        ("accumulator", C.POINTER(C.c_float)),
        ("framebuffer", C.POINTER(C.c_uint32)),
        ("pow_mode", C.c_int32),
        ("threads", C.c_int32),
        ("split", C.c_int32),
        ("rays_out", C.c_uint64),
        ("col_begin", C.c_int32),
        ("col_end", C.c_int32),
    ]


def build(force=False):
    """Compile the oracle with the committed Makefile (gcc -O2 -ffp-contract=off)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.srt_oracle_render.argtypes = [C.POINTER(Job)]
        L.srt_oracle_render.restype = C.c_int
        f3 = C.POINTER(C.c_float)
        L.srt_oracle_ray_direction.argtypes = [C.POINTER(Camera), C.c_int32, C.c_int32, C.c_int32, C.c_int32, f3]
        L.srt_oracle_ray_direction.restype = None
        L.srt_oracle_intersect.argtypes = [C.POINTER(Object), f3, f3, f3, f3, f3]
        L.srt_oracle_intersect.restype = C.c_int
        L.srt_oracle_closest.argtypes = [C.POINTER(Object), C.c_size_t, f3, f3, f3, f3, f3]
        L.srt_oracle_closest.restype = C.c_int
        L.srt_oracle_environment.argtypes = [C.POINTER(Environment), f3, C.c_int32, f3]
        L.srt_oracle_environment.restype = None
        L.srt_oracle_trace_sample.argtypes = [
            C.POINTER(Object), C.c_size_t, C.POINTER(Environment), C.POINTER(Camera),
            C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.c_int32, C.c_uint32, C.c_int32,
            f3, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
        ]
        L.srt_oracle_trace_sample.restype = None
        L.srt_oracle_tonemap_pack.argtypes = [f3]
        L.srt_oracle_tonemap_pack.restype = C.c_uint32
        L.srt_oracle_powf_shared.argtypes = [C.c_float, C.c_float]
        L.srt_oracle_powf_shared.restype = C.c_float
        L.srt_oracle_environment_default.argtypes = [C.POINTER(Environment)]
        L.srt_oracle_environment_default.restype = None
        _lib = L
    return _lib


# ----------------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------------
def f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def default_environment():
    e = Environment()
    lib().srt_oracle_environment_default(C.byref(e))
    return e


def default_camera(fov=55):
    """Camera as set up in Raytracer.cpp:295-297 (origin, identity basis), FOV :31."""
    c = Camera()
    c.position = f3((0, 0, 0))
    c.right = f3((1, 0, 0))
    c.up = f3((0, 1, 0))
    c.forward = f3((0, 0, 1))
    c.fov_degrees = fov
    return c


def make_objects(objs):
    """list of dicts -> ctypes array. Keys: type, position, radius, half_size, and the
    material fields smoothness, specular_amount, base, emissive, specular."""
    arr = (Object * max(1, len(objs)))()
    for i, o in enumerate(objs):
        a = arr[i]
        a.type = o.get("type", OBJ_NONE)
        a.position = f3(o.get("position", (0, 0, 0)))
        a.radius = float(o.get("radius", 0.0))
        a.half_size = f3(o.get("half_size", (0, 0, 0)))
        a.mesh = int(o.get("mesh", -1))
        a.material.smoothness = float(o.get("smoothness", 0.5))
        a.material.specular_amount = float(o.get("specular_amount", 0.0))
        a.material.base_color = f3(o.get("base", (1, 1, 1)))
        a.material.emissive_color = f3(o.get("emissive", (0, 0, 0)))
        a.material.specular_color = f3(o.get("specular", (1, 1, 1)))
    return arr, len(objs)


def make_meshes(meshes):
    """list of (vertices float32 [n,3], indices uint32 [m,3]) -> (ctypes Mesh array, count, keepalive)."""
    arr = (Mesh * max(1, len(meshes)))()
    keep = []
    for i, (v, ix) in enumerate(meshes):
        v = np.ascontiguousarray(v, dtype=np.float32).reshape(-1, 3)
        ix = np.ascontiguousarray(ix, dtype=np.uint32).reshape(-1, 3)
        keep += [v, ix]
        arr[i].vertices = v.ctypes.data_as(C.POINTER(C.c_float))
        arr[i].vertex_count = v.shape[0]
        arr[i].indices = ix.ctypes.data_as(C.POINTER(C.c_uint32))
        arr[i].triangle_count = ix.shape[0]
    return arr, len(meshes), keep


def uv_sphere(radius, stacks, slices):
    """Deterministic lat-long tessellation used by BASELINE config 4: `stacks` bands
    of latitude, `slices` of longitude; the pole bands are fans -> 2*slices*(stacks-1) triangles.
    float32 vertices, outward (counter-clockwise) winding.  Must match host/mesh.cpp."""
    vs = [(0.0, radius, 0.0)]
    import math
    for i in range(1, stacks):
        phi = math.pi * i / stacks
        y, r = radius * math.cos(phi), radius * math.sin(phi)
        for j in range(slices):
            th = 2.0 * math.pi * j / slices
            vs.append((r * math.cos(th), y, r * math.sin(th)))
    vs.append((0.0, -radius, 0.0))
    V = np.array(vs, dtype=np.float64).astype(np.float32)
    tri = []
    ring = lambda i, j: 1 + (i - 1) * slices + (j % slices)
    south = len(vs) - 1
    for j in range(slices):
        tri.append((0, ring(1, j + 1), ring(1, j)))
    for i in range(1, stacks - 1):
        for j in range(slices):
            a, b, c, d = ring(i, j), ring(i, j + 1), ring(i + 1, j), ring(i + 1, j + 1)
            tri.append((a, b, d))
            tri.append((a, d, c))
    for j in range(slices):
        tri.append((south, ring(stacks - 1, j), ring(stacks - 1, j + 1)))
    return V, np.array(tri, dtype=np.uint32)


def _clamp0(v):
    # Color's constructor clamps negatives (Common.hpp:253-257)
    return [0.0 if float(x) < 0 else float(x) for x in v]


def load_scene_json_py(path):
    """Independent reader of the reference scene format (Scene.hpp:27-80), python json.
    Returns a list of dicts for make_objects(). Mirrors: defaults at :61-68, missing
    "Material" -> Material() defaults (Common.hpp:313-318), unknown type -> inert."""
    with open(path, "r") as f:
        data = json.load(f)
    out = []
    for v in data["SceneObjects"]:
        r = v["Renderer"]
        o = {"position": v["Position"]}
        if r["Type"] == "Sphere":
            o["type"] = OBJ_SPHERE
            o["radius"] = r["Radius"]
        elif r["Type"] == "Cube":
            o["type"] = OBJ_BOX
            o["half_size"] = r["Size"]
        else:
            o["type"] = OBJ_NONE
        if "Material" in v:
            m = v["Material"]
            o["smoothness"] = m.get("Smoothness", 0.5)
            o["specular_amount"] = m.get("SpecularAmount", 0.1)
            o["specular"] = _clamp0(m.get("SpecularColor", [1, 1, 1]))
            o["base"] = _clamp0(m.get("Color", [1, 1, 1]))
            o["emissive"] = _clamp0(m.get("Emissive", [0, 0, 0]))
        out.append(o)
    return out


def render(objects, count, env, cam, width, height, *, spp=1, bounces=4, seed=0, first_sample=1,
           reset=True, rows=None, accumulator=None, pow_mode=POW_SHARED, threads=None,
           split=SPLIT_ROWS, preview=False, steps=1, stripe_width=0, selected=-1, meshes=None, cols=None):
    """Run the oracle. Returns (framebuffer uint32 [H,W] bottom-up, accumulator float32
    [H,W,4] scene rows, rays)."""
    if threads is None:
        threads = os.cpu_count() or 1
    if accumulator is None:
        accumulator = np.zeros((height, width, 4), dtype=np.float32)
    else:
        accumulator = np.ascontiguousarray(accumulator, dtype=np.float32).copy()
    fb = np.zeros((height, width), dtype=np.uint32)
    job = Job()
    job.objects = C.cast(objects, C.POINTER(Object))
    job.object_count = count
    if meshes is not None:
        job.meshes = C.cast(meshes[0], C.POINTER(Mesh))
        job.mesh_count = meshes[1]
    job.env = C.pointer(env)
    job.camera = C.pointer(cam)
    job.width, job.height = width, height
    rb, re = rows if rows is not None else (0, height)
    job.params = RenderParams(rb, re, first_sample, spp, bounces, seed,
                              (RENDER_RESET if reset else 0) | (RENDER_PREVIEW if preview else 0), steps, stripe_width, selected)
    job.accumulator = accumulator.ctypes.data_as(C.POINTER(C.c_float))
    job.framebuffer = fb.ctypes.data_as(C.POINTER(C.c_uint32))
    job.pow_mode = pow_mode
    job.threads = threads
    job.split = split
    if cols is not None:  # only this column window of the band (the rest of the returned arrays stays zero)
        job.col_begin, job.col_end = cols
    rc = lib().srt_oracle_render(C.byref(job))
    if rc != 0:
        raise RuntimeError("srt_oracle_render failed: %d" % rc)
    return fb, accumulator, int(job.rays_out)


def frame_hash(arr):
    """sha256 (first 16 hex digits) of the array's bytes — frame hash used by fixtures and bench."""
    import hashlib

    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()[:16]
