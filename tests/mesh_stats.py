#!/usr/bin/env python3
"""Development aid: traversal counters of the mesh kernel (builds libsrt_pathtrace_dev.so with STATS=1, or STATS=2 with SRT_STATS_MODE=2).
usage: python tests/mesh_stats.py [--mesh 224] [--spp 8]"""
import argparse, ctypes as C, importlib, json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--mesh", type=int, default=224)
ap.add_argument("--spp", type=int, default=8)
ap.add_argument("--bounces", type=int, default=8)
a = ap.parse_args()
srt = importlib.import_module("software-raytracer_amd")
srt.capi.use_dev_library(stats=int(os.environ.get("SRT_STATS_MODE", "1")))
L = srt.load_library()
path = os.path.join(ROOT, "software-raytracer_amd", "scenes", "Scene1.json")
sj = json.load(open(path))
sj["SceneObjects"][64]["Renderer"] = {"Type": "Mesh", "Primitive": "UVSphere", "Radius": 1.0, "Stacks": a.mesh, "Slices": a.mesh}
tmp = tempfile.NamedTemporaryFile("w", suffix=".json", delete=False); json.dump(sj, tmp); tmp.close()
sc = srt.host.Scene(tmp.name)
objs, n = sc.objects_copy(); meshes, nm = sc.meshes()
pt = srt.PathTracer(1920, 1080)
pt.set_meshes(meshes, nm); pt.set_scene(objs, n); pt.set_camera(srt.default_camera())
out = (C.c_ulonglong * 8)()
pt.render(spp=a.spp, bounces=a.bounces, seed=0, count_rays=True)
st = pt.stats()
L.srt_debug_read_stats(out)
pt.render(spp=a.spp, bounces=a.bounces, seed=0, count_rays=True)
st = pt.stats()
L.srt_debug_read_stats(out)
names = ["mesh phases (waves)", "go lanes", "node rounds", "node items", "leaf rounds", "leaf items", "overflows", "strict-mode entries"]
for k, v in zip(names, out): print("%-22s %d" % (k, v))
print("rays %d  kernel %.3f ms" % (st.rays, st.kernel_ms))
if os.environ.get("SRT_STATS_MODE") == "4":
    o = list(out)
    ph, nr, lr = max(o[6], 1), max(o[4], 1), max(o[5], 1)
    print("phases %d: %.0f cycles each | node rounds %.1f per phase, %.0f cycles each (%.0f until the node data is there) | leaf rounds %.1f per phase, %.0f cycles each | outside rounds %.0f cycles per phase" %
          (o[6], o[0] / ph, o[4] / ph, o[1] / nr, o[3] / nr, o[5] / ph, o[2] / lr, (o[0] - o[1] - o[2]) / ph))
    sys.exit(0)
if os.environ.get("SRT_STATS_MODE") == "2":
    print("histogram by rays entering the phase (1-2, 3-8, 9-32, 33-64): phases", list(out[:4]), "rounds", list(out[4:]))
    sys.exit(0)
g = out[1] or 1
print("per go-lane: node items %.1f leaf items %.1f | items/node round %.1f items/leaf round %.1f | rounds per phase %.1f" %
      (out[3] / g, out[5] / g, out[3] / max(out[2], 1), out[5] / max(out[4], 1), (out[2] + out[4]) / max(out[0], 1)))
