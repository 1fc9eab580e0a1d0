#!/usr/bin/env python3
"""Development aid: how much slower is the FIRST launch of a frame (no block costs recorded yet) than the steady state?
A fresh context is timed while the GPU is warm (another context has just rendered), so DVFS ramp-up is not billed.
usage: python tests/cold_bench.py [--mesh 224] [--spp 32] [--scene Scene1] [--dev]"""
import argparse, importlib, json, os, statistics, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="Scene1")
ap.add_argument("--spp", type=int, default=32)
ap.add_argument("--mesh", type=int, default=0)
ap.add_argument("--dev", action="store_true", help="use the development library (environment switches)")
ap.add_argument("--lib", default="", help="a library built elsewhere (A/B of two source versions on one box)")
a = ap.parse_args()
srt = importlib.import_module("software-raytracer_amd")
if a.lib:
    srt.capi._LIB = a.lib
elif a.dev:
    srt.capi.use_dev_library()
path = os.path.join(ROOT, "software-raytracer_amd", "scenes", a.scene + ".json")
if a.mesh:
    sj = json.load(open(path))
    sj["SceneObjects"][64]["Renderer"] = {"Type": "Mesh", "Primitive": "UVSphere", "Radius": 1.0, "Stacks": a.mesh, "Slices": a.mesh}
    tmp = tempfile.NamedTemporaryFile("w", suffix=".json", delete=False); json.dump(sj, tmp); tmp.close(); path = tmp.name
sc = srt.host.Scene(path)
objs, n = sc.objects_copy(); meshes, nm = sc.meshes()
def fresh():
    pt = srt.PathTracer(1920, 1080)
    pt.set_meshes(meshes, nm); pt.set_scene(objs, n); pt.set_camera(srt.default_camera())
    return pt
warm = fresh()
steady = []
for i in range(12):
    warm.render(spp=a.spp, bounces=8, seed=0); t = warm.stats().kernel_ms
    if i >= 4: steady.append(t)
first, second, third = [], [], []
for rep in range(5):
    warm.render(spp=a.spp, bounces=8, seed=0)        # keep the clocks up
    pt = fresh()
    pt.render(spp=1, bounces=1, seed=0, rows=(0, 8)); pt.wait()   # code-object load etc., like bench.py's priming launch
    warm.render(spp=a.spp, bounces=8, seed=0); warm.wait()
    ts = []
    for k in range(3):
        pt.render(spp=a.spp, bounces=8, seed=0); ts.append(pt.stats().kernel_ms)
    first.append(ts[0]); second.append(ts[1]); third.append(ts[2])
    pt.close()
m = statistics.median
print("%s%s spp %d: steady %.3f ms | fresh context: 1st launch %.3f (%+.1f %%), 2nd %.3f (%+.1f %%), 3rd %.3f (%+.1f %%)" %
      (a.scene, "+mesh%d" % a.mesh if a.mesh else "", a.spp, m(steady), m(first), 100 * (m(first) / m(steady) - 1), m(second),
       100 * (m(second) / m(steady) - 1), m(third), 100 * (m(third) / m(steady) - 1)))
