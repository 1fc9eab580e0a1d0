"""ctypes bindings of include/srt_pathtrace.h (libsrt_pathtrace.so).  Plumbing only.

Every call goes through the C-ABI — the same entry points a C, C++ or FFI caller would
bind.  No fallback: a missing library raises ImportError-like SrtError at load time,
a missing GPU makes srt_create fail with SRT_ERR_NO_DEVICE.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_PKG, "libsrt_pathtrace.so")

OK, ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_STATE, ERR_OOM = range(6)
OBJ_NONE, OBJ_SPHERE, OBJ_BOX, OBJ_MESH = 0, 1, 2, 3
RENDER_RESET, RENDER_COUNT_RAYS, RENDER_PREVIEW, RENDER_COUNT_WORK, RENDER_NO_TIMING = 1, 2, 4, 8, 16
ABI_VERSION = 6

# every symbol include/srt_pathtrace.h declares (tests check the library exports them all)
EXPORTS = [
    "srt_abi_version", "srt_device_count", "srt_create", "srt_destroy", "srt_last_error",
    "srt_set_scene", "srt_set_meshes", "srt_set_environment", "srt_environment_default", "srt_set_camera",
    "srt_set_stream", "srt_bind_output", "srt_device_framebuffer", "srt_device_accumulator",
    "srt_render", "srt_wait", "srt_poll", "srt_get_stats", "srt_get_work_counts", "srt_pick", "srt_read_framebuffer",
    "srt_read_framebuffer_async", "srt_read_accumulator", "srt_write_accumulator", "srt_gather_band", "srt_gather_path", "srt_estimate_row_costs",
    "srt_selftest_arith",
]


class SrtError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("srt error %d: %s" % (code, text))
        self.code = code


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


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("path_samples", C.c_uint64), ("kernel_ms", C.c_float), ("sample_chunks", C.c_uint32),
                ("tile_rows", C.c_uint32), ("chunk_samples", C.c_uint32), ("shape_source", C.c_uint32)]


class WorkCounts(C.Structure):
    """srt_work_counts: what one render's kernels executed (SRT_RENDER_COUNT_WORK)."""
    _fields_ = [("valid", C.c_uint32), ("reserved", C.c_uint32)] + [(n, C.c_uint64) for n in (
        "waves", "pool_steps", "closest_hit_calls", "uniform_sphere_tests", "cluster_bound_tests", "cluster_sphere_tests",
        "cluster_items", "box_tests", "bvh_child_tests", "triangle_tests", "bvh_node_rounds", "mesh_phases")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_ if n != "reserved"}


def lib_path():
    return _LIB


def build_native(force=False):
    """hipcc --offload-arch=gfx950 build of the C-ABI library (cross-compiles without a GPU)."""
    args = ["make", "-C", os.path.join(_PKG, "csrc"), "-s"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return _LIB


_lib = None


def use_dev_library(stats=None, rebuild=False):
    """Development tools only (tests/ab_bench.py, tests/mesh_stats.py, tools/): build and select
    libsrt_pathtrace_dev.so (-DSRT_DEV: environment switches, srt_debug_* entry points, optional STATS
    counters).  Must be called before the first load_library(); the product and the tests never call it."""
    global _LIB
    assert _lib is None, "use_dev_library() must come before load_library()"
    dev = os.path.join(_PKG, "libsrt_pathtrace_dev%s.so" % ("_stats%d" % stats if stats is not None else ""))
    args = ["make", "-C", os.path.join(_PKG, "csrc"), "-s", "dev"]
    if stats is not None:
        args.append("STATS=%d" % stats)
    if rebuild:
        args.append("-B")
    subprocess.check_call(args)
    _LIB = dev
    return dev


def load_library():
    """dlopen libsrt_pathtrace.so and declare prototypes. Raises SrtError if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise SrtError(ERR_STATE, "%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                  "or `make -C software-raytracer_amd/csrc` (there is no fallback path)" % _LIB)
    _lib = open_library(_LIB)
    return _lib


def open_library(path):
    """dlopen one build of the C-ABI library and declare its prototypes (load_library() for the product's; development
    tools open several builds side by side for interleaved A/B timing, tests/ab_libs.py)."""
    L = C.CDLL(path)
    ctx = C.c_void_p
    L.srt_abi_version.restype = C.c_int
    L.srt_device_count.argtypes = [C.POINTER(C.c_int)]
    L.srt_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(ctx)]
    L.srt_destroy.argtypes = [ctx]
    L.srt_last_error.argtypes = [ctx]
    L.srt_last_error.restype = C.c_char_p
    L.srt_set_scene.argtypes = [ctx, C.POINTER(Object), C.c_size_t]
    L.srt_set_meshes.argtypes = [ctx, C.POINTER(Mesh), C.c_size_t]
    L.srt_set_environment.argtypes = [ctx, C.POINTER(Environment)]
    L.srt_environment_default.argtypes = [C.POINTER(Environment)]
    L.srt_set_camera.argtypes = [ctx, C.POINTER(Camera)]
    L.srt_set_stream.argtypes = [ctx, C.c_void_p]
    L.srt_bind_output.argtypes = [ctx, C.c_void_p, C.c_void_p]
    L.srt_device_framebuffer.argtypes = [ctx, C.POINTER(C.c_void_p)]
    L.srt_device_accumulator.argtypes = [ctx, C.POINTER(C.c_void_p)]
    L.srt_render.argtypes = [ctx, C.POINTER(RenderParams)]
    L.srt_wait.argtypes = [ctx]
    L.srt_poll.argtypes = [ctx, C.POINTER(C.c_int)]
    L.srt_get_stats.argtypes = [ctx, C.POINTER(Stats)]
    L.srt_get_work_counts.argtypes = [ctx, C.POINTER(WorkCounts)]
    L.srt_pick.argtypes = [ctx, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.srt_read_framebuffer.argtypes = [ctx, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
    L.srt_read_framebuffer_async.argtypes = [ctx, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
    L.srt_read_accumulator.argtypes = [ctx, C.POINTER(C.c_float)]
    L.srt_write_accumulator.argtypes = [ctx, C.POINTER(C.c_float)]
    L.srt_gather_band.argtypes = [ctx, ctx, C.c_int, C.c_int]
    L.srt_gather_path.argtypes = [ctx]
    L.srt_gather_path.restype = C.c_char_p
    L.srt_estimate_row_costs.argtypes = [ctx, C.c_int, C.c_uint32, C.POINTER(C.c_float)]
    L.srt_selftest_arith.argtypes = [C.c_int, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64)]
    for name in EXPORTS:
        fn = getattr(L, name)
        if name not in ("srt_last_error", "srt_gather_path"):
            fn.restype = C.c_int
    return L


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def default_environment():
    e = Environment()
    rc = load_library().srt_environment_default(C.byref(e))
    if rc:
        raise SrtError(rc, "srt_environment_default")
    return e


def default_camera(fov=55):
    """Raytracer.cpp:295-297 (origin, identity basis) and FOV :31."""
    c = Camera()
    c.position = _f3((0, 0, 0))
    c.right = _f3((1, 0, 0))
    c.up = _f3((0, 1, 0))
    c.forward = _f3((0, 0, 1))
    c.fov_degrees = fov
    return c


class PathTracer:
    """Thin RAII wrapper of an srt_context handle."""

    def __init__(self, width, height, device=0, lib=None):
        self.L = lib if lib is not None else load_library()
        self.width, self.height = int(width), int(height)
        self._h = C.c_void_p()
        rc = self.L.srt_create(int(device), self.width, self.height, C.byref(self._h))
        if rc:
            raise SrtError(rc, (self.L.srt_last_error(None) or b"").decode())

    def _ck(self, rc):
        if rc:
            raise SrtError(rc, (self.L.srt_last_error(self._h) or b"").decode())

    def close(self):
        if self._h:
            self.L.srt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- state ---------------------------------------------------------------------
    def set_scene(self, objects, count=None):
        n = len(objects) if count is None else count
        ptr = C.cast(objects, C.POINTER(Object)) if n else None
        self._ck(self.L.srt_set_scene(self._h, ptr, n))

    def set_meshes(self, meshes, count=None):
        """EXTENSION: geometry for SRT_OBJ_MESH objects; call before set_scene."""
        n = len(meshes) if count is None else count
        self._ck(self.L.srt_set_meshes(self._h, C.cast(meshes, C.POINTER(Mesh)) if n else None, n))

    def set_environment(self, env):
        self._ck(self.L.srt_set_environment(self._h, C.byref(env)))

    def set_camera(self, cam):
        self._ck(self.L.srt_set_camera(self._h, C.byref(cam)))

    def set_stream(self, stream_ptr):
        self._ck(self.L.srt_set_stream(self._h, C.c_void_p(stream_ptr)))

    def bind_output(self, d_framebuffer=None, d_accumulator=None):
        self._ck(self.L.srt_bind_output(self._h, C.c_void_p(d_framebuffer or 0), C.c_void_p(d_accumulator or 0)))

    # ---- hot path --------------------------------------------------------------------
    def render(self, *, spp=1, bounces=4, seed=0, first_sample=1, reset=True, rows=None, count_rays=False,
               preview=False, steps=1, stripe_width=0, selected=-1, count_work=False, timing=True):
        rb, re = rows if rows is not None else (0, self.height)
        flags = ((RENDER_RESET if reset else 0) | (RENDER_COUNT_RAYS if count_rays else 0) | (RENDER_PREVIEW if preview else 0) |
                 (RENDER_COUNT_WORK if count_work else 0) | (0 if timing else RENDER_NO_TIMING))
        p = RenderParams(rb, re, first_sample, spp, bounces, seed, flags, steps, stripe_width, selected)
        self._ck(self.L.srt_render(self._h, C.byref(p)))

    def wait(self):
        self._ck(self.L.srt_wait(self._h))

    def poll(self):
        d = C.c_int(0)
        self._ck(self.L.srt_poll(self._h, C.byref(d)))
        return bool(d.value)

    def pick(self, x, y):
        """Raytracer.cpp:525-541; y in scene rows. Returns the list index or -1."""
        idx = C.c_int(-2)
        self._ck(self.L.srt_pick(self._h, int(x), int(y), C.byref(idx)))
        return idx.value

    def stats(self):
        s = Stats()
        self._ck(self.L.srt_get_stats(self._h, C.byref(s)))
        return s

    def work_counts(self):
        """srt_get_work_counts: the loop counts of the last render (it must have had count_work=True)."""
        w = WorkCounts()
        self._ck(self.L.srt_get_work_counts(self._h, C.byref(w)))
        return w

    # ---- buffers ---------------------------------------------------------------------
    def framebuffer(self, rows=None):
        rb, re = rows if rows is not None else (0, self.height)
        out = np.empty((re - rb, self.width), dtype=np.uint32)
        self._ck(self.L.srt_read_framebuffer(self._h, out.ctypes.data_as(C.c_void_p), self.width * 4, rb, re))
        return out

    def read_framebuffer_async(self, dst_ptr, rows=None, copy_stream=0):
        """srt_read_framebuffer_async: memory rows `rows` into host memory at dst_ptr (pinned, tightly packed), enqueued on
        hipStream_t `copy_stream` (0: the launch stream) behind the renders enqueued so far; returns at once."""
        rb, re = rows if rows is not None else (0, self.height)
        self._ck(self.L.srt_read_framebuffer_async(self._h, C.c_void_p(dst_ptr), self.width * 4, rb, re, C.c_void_p(copy_stream or 0)))

    def accumulator(self):
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._ck(self.L.srt_read_accumulator(self._h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def write_accumulator(self, arr):
        a = np.ascontiguousarray(arr, dtype=np.float32)
        assert a.shape == (self.height, self.width, 4)
        self._ck(self.L.srt_write_accumulator(self._h, a.ctypes.data_as(C.POINTER(C.c_float))))

    def estimate_row_costs(self, bounces, seed=0):
        """srt_estimate_row_costs: relative cost per memory row (list of floats) from the device-side probe."""
        out = (C.c_float * self.height)()
        self._ck(self.L.srt_estimate_row_costs(self._h, int(bounces), int(seed), out))
        return list(out)

    def gather_band_from(self, src, rows):
        """srt_gather_band: memory rows `rows` of PathTracer `src`'s framebuffer into this one's (device to device)."""
        self._ck(self.L.srt_gather_band(self._h, src._h, int(rows[0]), int(rows[1])))

    def gather_path(self):
        """srt_gather_path: which way this tracer's last band went in gather_band_from (text)."""
        return (self.L.srt_gather_path(self._h) or b"").decode()

    def device_framebuffer_ptr(self):
        p = C.c_void_p()
        self._ck(self.L.srt_device_framebuffer(self._h, C.byref(p)))
        return p.value
