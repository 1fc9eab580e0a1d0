"""ctypes bindings of the C++ host library (libsrt_host.so): scene JSON reader/writer in
the reference's format (Raytracer/Scene.hpp), Transform, progressive renderer.  Plumbing
only — the logic lives in software-raytracer_amd/host/*.cpp."""
import ctypes as C
import os
import subprocess

import numpy as np

from .capi import Mesh, Object, Stats

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_PKG, "libsrt_host.so")

EXPORTS = [
    "srt_host_scene_load", "srt_host_scene_new", "srt_host_scene_free", "srt_host_scene_count",
    "srt_host_scene_objects", "srt_host_scene_mesh_count", "srt_host_scene_mesh", "srt_host_scene_error", "srt_host_scene_name", "srt_host_scene_object_name",
    "srt_host_scene_add", "srt_host_scene_remove", "srt_host_scene_save_as", "srt_host_scene_dump",
    "srt_host_format_double", "srt_host_json_roundtrip", "srt_host_rotate_about_axis", "srt_host_last_error",
    "srt_host_renderer_create", "srt_host_renderer_destroy", "srt_host_renderer_set_scene",
    "srt_host_renderer_set_band", "srt_host_renderer_settings", "srt_host_renderer_set_camera",
    "srt_host_renderer_invalidate", "srt_host_renderer_mode", "srt_host_renderer_pick", "srt_host_renderer_render_frame", "srt_host_renderer_render_samples",
    "srt_host_renderer_accumulation_frames", "srt_host_renderer_wait", "srt_host_renderer_read_framebuffer",
    "srt_host_renderer_read_accumulator", "srt_host_renderer_stats", "srt_host_renderer_handle",
    "srt_host_multi_create", "srt_host_multi_destroy", "srt_host_multi_set_scene", "srt_host_multi_configure",
    "srt_host_multi_render_samples", "srt_host_multi_read_framebuffer", "srt_host_multi_band", "srt_host_multi_stats", "srt_host_multi_balance", "srt_host_multi_use_equal_bands",
    "srt_host_multi_use_manual_bands", "srt_host_multi_set_auto_balance_min_samples", "srt_host_multi_set_row_band",
]

_lib = None


def build_native(force=False):
    args = ["make", "-C", os.path.join(_PKG, "host"), "-s"] + (["-B"] if force else [])
    subprocess.check_call(args)
    return _LIB


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise RuntimeError("%s not built: make -C software-raytracer_amd/host" % _LIB)
    L = C.CDLL(_LIB)
    vp = C.c_void_p
    L.srt_host_scene_load.argtypes = [C.c_char_p]
    L.srt_host_scene_load.restype = vp
    L.srt_host_scene_new.argtypes = [C.c_char_p]
    L.srt_host_scene_new.restype = vp
    L.srt_host_scene_free.argtypes = [vp]
    L.srt_host_scene_free.restype = None
    L.srt_host_scene_count.argtypes = [vp]
    L.srt_host_scene_count.restype = C.c_size_t
    L.srt_host_scene_objects.argtypes = [vp]
    L.srt_host_scene_objects.restype = C.POINTER(Object)
    L.srt_host_scene_mesh_count.argtypes = [vp]
    L.srt_host_scene_mesh_count.restype = C.c_size_t
    L.srt_host_scene_mesh.argtypes = [vp, C.c_size_t, C.POINTER(Mesh)]
    for n in ("srt_host_scene_error", "srt_host_scene_name", "srt_host_scene_dump"):
        getattr(L, n).argtypes = [vp]
        getattr(L, n).restype = C.c_char_p
    L.srt_host_scene_object_name.argtypes = [vp, C.c_size_t]
    L.srt_host_scene_object_name.restype = C.c_char_p
    L.srt_host_scene_add.argtypes = [vp, C.POINTER(Object), C.c_char_p]
    L.srt_host_scene_add.restype = None
    L.srt_host_scene_remove.argtypes = [vp, C.c_size_t]
    L.srt_host_scene_remove.restype = C.c_int
    L.srt_host_scene_save_as.argtypes = [vp, C.c_char_p]
    L.srt_host_scene_save_as.restype = None
    L.srt_host_format_double.argtypes = [C.c_double, C.c_char_p, C.c_size_t]
    L.srt_host_format_double.restype = C.c_size_t
    L.srt_host_json_roundtrip.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]
    L.srt_host_json_roundtrip.restype = C.c_size_t
    L.srt_host_rotate_about_axis.argtypes = [C.POINTER(C.c_float), C.c_float, C.POINTER(C.c_float)]
    L.srt_host_rotate_about_axis.restype = None
    L.srt_host_last_error.restype = C.c_char_p
    L.srt_host_renderer_create.argtypes = [C.c_int, C.c_int, C.c_int]
    L.srt_host_renderer_create.restype = vp
    L.srt_host_renderer_destroy.argtypes = [vp]
    L.srt_host_renderer_destroy.restype = None
    L.srt_host_renderer_set_scene.argtypes = [vp, vp]
    L.srt_host_renderer_set_band.argtypes = [vp, C.c_int, C.c_int]
    L.srt_host_renderer_settings.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_uint32]
    L.srt_host_renderer_set_camera.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.srt_host_renderer_mode.argtypes = [vp, C.c_int, C.c_float, C.c_int]
    L.srt_host_renderer_pick.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.srt_host_renderer_invalidate.argtypes = [vp]
    L.srt_host_renderer_invalidate.restype = None
    L.srt_host_renderer_render_frame.argtypes = [vp]
    L.srt_host_renderer_render_samples.argtypes = [vp, C.c_uint32, C.c_int]
    L.srt_host_renderer_accumulation_frames.argtypes = [vp]
    L.srt_host_renderer_wait.argtypes = [vp]
    L.srt_host_renderer_read_framebuffer.argtypes = [vp, vp, C.c_size_t]
    L.srt_host_renderer_read_accumulator.argtypes = [vp, C.POINTER(C.c_float)]
    L.srt_host_renderer_stats.argtypes = [vp, C.POINTER(Stats)]
    L.srt_host_renderer_handle.argtypes = [vp]
    L.srt_host_multi_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int]
    L.srt_host_multi_create.restype = vp
    L.srt_host_multi_destroy.argtypes = [vp]
    L.srt_host_multi_destroy.restype = None
    L.srt_host_multi_set_scene.argtypes = [vp, vp]
    L.srt_host_multi_configure.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int, C.c_uint32]
    L.srt_host_multi_render_samples.argtypes = [vp, C.c_uint32, C.c_int]
    L.srt_host_multi_read_framebuffer.argtypes = [vp, vp, C.c_size_t]
    L.srt_host_multi_band.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.srt_host_multi_stats.argtypes = [vp, C.POINTER(Stats), C.c_int]
    L.srt_host_multi_balance.argtypes = [vp]
    L.srt_host_multi_use_equal_bands.argtypes = [vp, C.c_int]
    L.srt_host_multi_use_manual_bands.argtypes = [vp, C.c_int]
    L.srt_host_multi_set_auto_balance_min_samples.argtypes = [vp, C.c_uint32]
    L.srt_host_multi_set_row_band.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.srt_host_renderer_handle.restype = vp
    _lib = L
    return L


class Scene:
    """Scene(file) of Raytracer/Scene.hpp through the C++ host mirror."""

    def __init__(self, path, load=True):
        self.L = load_library()
        self.path = path
        self._h = (self.L.srt_host_scene_load if load else self.L.srt_host_scene_new)(path.encode())
        if not self._h:
            raise MemoryError("srt_host_scene")

    def close(self):
        if self._h:
            self.L.srt_host_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return self.L.srt_host_scene_count(self._h)

    @property
    def error(self):
        return self.L.srt_host_scene_error(self._h).decode()

    @property
    def name(self):
        return self.L.srt_host_scene_name(self._h).decode()

    def object_name(self, i):
        return self.L.srt_host_scene_object_name(self._h, i).decode()

    def objects(self):
        """(ctypes pointer to srt_object[], count) — ObjectsToRender in list order."""
        return self.L.srt_host_scene_objects(self._h), len(self)

    def objects_copy(self):
        ptr, n = self.objects()
        arr = (Object * max(n, 1))()
        for i in range(n):
            arr[i] = ptr[i]
        return arr, n

    def meshes(self):
        """EXTENSION: (ctypes Mesh array, count) of the scene's "Mesh" renderers (pointers into the scene)."""
        n = self.L.srt_host_scene_mesh_count(self._h)
        arr = (Mesh * max(n, 1))()
        for i in range(n):
            self.L.srt_host_scene_mesh(self._h, i, C.byref(arr[i]))
        return arr, n

    def add(self, obj, name=""):
        self.L.srt_host_scene_add(self._h, C.byref(obj), name.encode())

    def remove(self, index):
        return bool(self.L.srt_host_scene_remove(self._h, index))

    def save_as(self, path):
        self.L.srt_host_scene_save_as(self._h, path.encode())

    def dump(self):
        return self.L.srt_host_scene_dump(self._h).decode()


def format_double(v):
    buf = C.create_string_buffer(64)
    load_library().srt_host_format_double(float(v), buf, 64)
    return buf.value.decode()


def json_roundtrip(text, indent=4):
    """Json::parse + dump(indent) of the host JSON code; None on a parse error."""
    data = text.encode() if isinstance(text, str) else bytes(text)
    cap = 8 * len(data) + 1024
    buf = C.create_string_buffer(cap)
    n = load_library().srt_host_json_roundtrip(data, indent, buf, cap)
    if n == C.c_size_t(-1).value:
        return None
    return buf.value.decode("utf-8", "surrogateescape")


def rotate_about_axis(basis9, angle, axis):
    b = (C.c_float * 9)(*[float(x) for x in basis9])
    a = (C.c_float * 3)(*[float(x) for x in axis])
    load_library().srt_host_rotate_about_axis(b, float(angle), a)
    return list(b)


class Renderer:
    """PathTraceRenderer (host/renderer.hpp): camera, settings, progressive state."""

    def __init__(self, width, height, device=0):
        self.L = load_library()
        self.width, self.height = width, height
        self._band = (0, height)
        self._h = self.L.srt_host_renderer_create(device, width, height)
        if not self._h:
            raise RuntimeError(self.L.srt_host_last_error().decode())

    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError("host renderer error %d: %s" % (rc, self.L.srt_host_last_error().decode()))

    def close(self):
        if self._h:
            self.L.srt_host_renderer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_scene(self, scene):
        self._ck(self.L.srt_host_renderer_set_scene(self._h, scene._h))

    def set_band(self, rb, re):
        self._ck(self.L.srt_host_renderer_set_band(self._h, rb, re))
        self._band = (rb, re)

    def settings(self, fov=55, max_bounces=2, target_frames=4096, seed=0):
        self._ck(self.L.srt_host_renderer_settings(self._h, fov, max_bounces, target_frames, seed))

    def set_camera(self, position, basis9):
        p = (C.c_float * 3)(*[float(x) for x in position])
        b = (C.c_float * 9)(*[float(x) for x in basis9])
        self._ck(self.L.srt_host_renderer_set_camera(self._h, p, b))

    def mode(self, simpledraw=True, screen_scale=0.5, selected=-1):
        self._ck(self.L.srt_host_renderer_mode(self._h, 1 if simpledraw else 0, float(screen_scale), int(selected)))

    def pick(self, mouse_x, mouse_y):
        idx = C.c_int(-2)
        self._ck(self.L.srt_host_renderer_pick(self._h, mouse_x, mouse_y, C.byref(idx)))
        return idx.value

    def invalidate(self):
        self.L.srt_host_renderer_invalidate(self._h)

    def render_frame(self):
        rc = self.L.srt_host_renderer_render_frame(self._h)
        if rc < 0:
            raise RuntimeError(self.L.srt_host_last_error().decode())
        return bool(rc)

    def render_samples(self, count, count_rays=False):
        self._ck(self.L.srt_host_renderer_render_samples(self._h, count, 1 if count_rays else 0))

    @property
    def accumulation_frames(self):
        return self.L.srt_host_renderer_accumulation_frames(self._h)

    def wait(self):
        self._ck(self.L.srt_host_renderer_wait(self._h))

    def framebuffer(self):
        rb, re = self._band
        out = np.empty((re - rb, self.width), dtype=np.uint32)
        self._ck(self.L.srt_host_renderer_read_framebuffer(self._h, out.ctypes.data_as(C.c_void_p), self.width * 4))
        return out

    def accumulator(self):
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._ck(self.L.srt_host_renderer_read_accumulator(self._h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def stats(self):
        s = Stats()
        self._ck(self.L.srt_host_renderer_stats(self._h, C.byref(s)))
        return s

    def handle(self):
        return self.L.srt_host_renderer_handle(self._h)


class MultiRenderer:
    """MultiGpuRenderer (host/renderer.hpp): one frame over several devices of one node in ONE process — equal
    memory-row bands, one context and stream per device, joined by srt_gather_band into the first device's
    framebuffer.  A device may be listed several times (how it is tested on a one-GPU box)."""

    def __init__(self, devices, width, height):
        self.L = load_library()
        self.width, self.height, self.n = width, height, len(devices)
        arr = (C.c_int * len(devices))(*[int(d) for d in devices])
        self._h = self.L.srt_host_multi_create(arr, len(devices), width, height)
        if not self._h:
            raise RuntimeError(self.L.srt_host_last_error().decode())

    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError("host multi-renderer error %d: %s" % (rc, self.L.srt_host_last_error().decode()))

    def close(self):
        if self._h:
            self.L.srt_host_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_scene(self, scene):
        self._ck(self.L.srt_host_multi_set_scene(self._h, scene._h))

    def configure(self, position=(0, 0, 0), basis9=(1, 0, 0, 0, 1, 0, 0, 0, 1), fov=55, max_bounces=2, seed=0):
        p = (C.c_float * 3)(*[float(x) for x in position])
        b = (C.c_float * 9)(*[float(x) for x in basis9])
        self._ck(self.L.srt_host_multi_configure(self._h, p, b, int(fov), int(max_bounces), seed))

    def render_samples(self, count, count_rays=False):
        self._ck(self.L.srt_host_multi_render_samples(self._h, count, 1 if count_rays else 0))

    def framebuffer(self):
        out = np.empty((self.height, self.width), dtype=np.uint32)
        self._ck(self.L.srt_host_multi_read_framebuffer(self._h, out.ctypes.data_as(C.c_void_p), self.width * 4))
        return out

    def balance_bands(self):
        """Bands of equal estimated cost (srt_estimate_row_costs) instead of equal height."""
        self._ck(self.L.srt_host_multi_balance(self._h))

    def use_equal_bands(self, equal=True):
        """north_star's literal equal bands instead of the default split (bands of equal estimated cost, made by the first
        render_samples after the scene / camera / bounces change whose count is at least 32 per device)."""
        self._ck(self.L.srt_host_multi_use_equal_bands(self._h, 1 if equal else 0))

    def use_manual_bands(self, manual=True):
        """Leave the bands set through set_row_band() alone (without this the automatic split replaces them)."""
        self._ck(self.L.srt_host_multi_use_manual_bands(self._h, 1 if manual else 0))

    def set_row_band(self, i, begin, end):
        self._ck(self.L.srt_host_multi_set_row_band(self._h, int(i), int(begin), int(end)))

    def set_auto_balance_min_samples(self, per_device):
        """The automatic split probes only for requests of at least this many samples per device (default 32)."""
        self._ck(self.L.srt_host_multi_set_auto_balance_min_samples(self._h, int(per_device)))

    def band(self, i):
        b, e = C.c_int(), C.c_int()
        self._ck(self.L.srt_host_multi_band(self._h, int(i), C.byref(b), C.byref(e)))
        return b.value, e.value

    def stats(self):
        arr = (Stats * self.n)()
        self._ck(self.L.srt_host_multi_stats(self._h, arr, self.n))
        return list(arr)
