#!/usr/bin/env python3
"""Development aid: kernel time of one row band at a high sample count (a rank's share of BASELINE config 3 / 5), with the
development library so that SRT_DEFER (samples per chunk) and SRT_TILE_H can be varied from the environment.
usage: SRT_DEFER=32 python tests/band_bench.py [--rows 540,675] [--spp 512] [--mesh 0] [--width 1920 --height 1080] [--bounces 8]"""
import argparse, importlib, json, os, statistics, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--rows", default="540,675")
ap.add_argument("--spp", type=int, default=512)
ap.add_argument("--mesh", type=int, default=0)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--bounces", type=int, default=8)
ap.add_argument("--rounds", type=int, default=6)
a = ap.parse_args()
srt = importlib.import_module("software-raytracer_amd")
srt.capi.use_dev_library()
path = os.path.join(ROOT, "software-raytracer_amd", "scenes", "Scene1.json")
if a.mesh:
    sj = json.load(open(path))
    sj["SceneObjects"][64]["Renderer"] = {"Type": "Mesh", "Primitive": "UVSphere", "Radius": 1.0, "Stacks": a.mesh, "Slices": a.mesh}
    tmp = tempfile.NamedTemporaryFile("w", suffix=".json", delete=False); json.dump(sj, tmp); tmp.close(); path = tmp.name
sc = srt.host.Scene(path)
objs, n = sc.objects_copy(); meshes, nm = sc.meshes()
pt = srt.PathTracer(a.width, a.height)
pt.set_meshes(meshes, nm); pt.set_scene(objs, n); pt.set_camera(srt.default_camera())
rows = tuple(int(v) for v in a.rows.split(","))
ts = []
for i in range(a.rounds + 2):
    pt.render(spp=a.spp, bounces=a.bounces, seed=0, rows=rows)
    st = pt.stats()
    ts.append(st.kernel_ms)
print("SRT_DEFER=%s rows %s spp %d: median %.3f ms (min %.3f), %d sample chunks" % (os.environ.get("SRT_DEFER", "-"), a.rows, a.spp, statistics.median(ts[2:]), min(ts[2:]), st.sample_chunks))
