"""software-raytracer_amd — MI355X-native drop-in for ONE hot path of
JoshuaLim007/Software-Raytracer: the per-pixel trace / shade / accumulate loop
(Raytracer/Raytracer.cpp:63-257).

The product is native: `libsrt_pathtrace.so` (C-ABI in include/srt_pathtrace.h + the
hand-written gfx950 kernel) and `libsrt_host.so` (C++ host mirror: scene JSON, camera,
progressive accumulator).  This Python package is plumbing for tests and bench.py:
ctypes bindings over those two libraries.  It contains no compute and no CPU fallback —
if the HIP library is missing or no GPU is present, calls fail loudly.

The directory name has a hyphen, so import it with
    importlib.import_module("software-raytracer_amd")
"""
# torch bundles its own libamdhip64 (same soname as /opt/rocm's).  Import it before any native
# library of this package is dlopen-ed so that the process ends up with ONE HIP runtime that
# torch tensors, streams and our kernels share.  (The C++ host / CLI never involve torch.)
try:
    import torch  # noqa: F401
except ImportError:  # plumbing only; the native libraries work without it
    pass

from . import capi  # noqa: F401
from .capi import (  # noqa: F401
    Camera,
    Environment,
    Material,
    Mesh,
    Object,
    PathTracer,
    RenderParams,
    SrtError,
    Stats,
    build_native,
    default_camera,
    default_environment,
    lib_path,
    load_library,
)
from . import host  # noqa: F401

__all__ = [
    "capi", "host", "PathTracer", "SrtError", "Object", "Material", "Camera", "Environment",
    "RenderParams", "Stats", "default_camera", "default_environment", "build_native", "lib_path",
    "load_library",
]
