"""ctypes bindings of the C++ host library (libsrt_host.so): scene JSON reader/writer in
the reference's format (Raytracer/Scene.hpp), camera Transform, progressive renderer.
Plumbing only; filled in by software-raytracer_amd/host/ (C++)."""
