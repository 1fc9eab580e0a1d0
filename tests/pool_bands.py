#!/usr/bin/env python3
"""Development aid: for row bands of a config, the production kernel time next to the path pool's counters (STATS=4 library):
does  time ~ a * steps + b * exact rounds + c * pixels  hold?  usage: python tests/pool_bands.py <3|5> [--bands a:b,c:d]"""
import argparse, ctypes as C, importlib, json, os, statistics, sys, tempfile
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("config", type=int)
ap.add_argument("--bands", default="")
ap.add_argument("--json", default="")
a = ap.parse_args()
CFG = {3: ("Scene1", 0, 1920, 1080, 512, 8), 5: ("Scene1", 224, 3840, 2160, 1024, 16), 6: ("Scene_indirect", 0, 1920, 1080, 512, 8), 7: ("Scene3", 0, 1920, 1080, 512, 8)}
scene, mesh, W, H, spp, bounces = CFG[a.config]
srt = importlib.import_module("software-raytracer_amd")
prod = srt.capi.open_library(os.path.join(ROOT, "software-raytracer_amd", "libsrt_pathtrace.so"))
st4 = srt.capi.open_library(os.path.join(ROOT, "software-raytracer_amd", "libsrt_pathtrace_dev_stats4.so"))
st4.srt_debug_read_stats.argtypes = [C.POINTER(C.c_ulonglong)]
path = os.path.join(ROOT, "software-raytracer_amd", "scenes", scene + ".json")
if mesh:
    sj = json.load(open(path))
    sj["SceneObjects"][64]["Renderer"] = {"Type": "Mesh", "Primitive": "UVSphere", "Radius": 1.0, "Stacks": mesh, "Slices": mesh}
    tmp = tempfile.NamedTemporaryFile("w", suffix=".json", delete=False); json.dump(sj, tmp); tmp.close(); path = tmp.name
sc = srt.host.Scene(path)
objs, n = sc.objects_copy(); meshes, nm = sc.meshes()
if a.bands:
    bands = [tuple(int(v) for v in b.split(":")) for b in a.bands.split(",")]
else:
    h = H // 8
    bands = [(k * h, (k + 1) * h) for k in range(8)] + [(k * h + h // 2, (k + 1) * h + h // 2) for k in range(7)]
doc = []
for rows in bands:
    res = {}
    for name, lib in (("prod", prod), ("st4", st4)):
        pt = srt.PathTracer(W, H, lib=lib)
        pt.set_meshes(meshes, nm); pt.set_scene(objs, n); pt.set_camera(srt.default_camera())
        ts = []
        out = (C.c_ulonglong * 8)()
        for i in range(6 if name == "prod" else 3):
            pt.render(spp=spp, bounces=bounces, seed=0, rows=rows, count_rays=(name == "st4"))
            s = pt.stats(); ts.append(s.kernel_ms)
            if name == "st4":
                st4.srt_debug_read_stats(out)
        res[name] = statistics.median(ts[2:])
        res["chunks"] = int(s.sample_chunks)
        if name == "st4":
            res["stats"] = list(out); res["rays"] = int(s.rays)
        pt.close()
    o = res["stats"]
    print("%5d-%5d prod %7.3f ms chunks %2d | steps %9d busy/step %.1f items/step %.1f rounds/step %.2f folds/step %.2f rays %d" %
          (rows[0], rows[1], res["prod"], res["chunks"], o[0], o[1] / max(o[0], 1), o[2] / max(o[0], 1), o[3] / max(o[0], 1), o[4] / max(o[0], 1), res["rays"]), flush=True)
    doc.append({"rows": rows, **res})
if a.json:
    json.dump(doc, open(a.json, "w"))
