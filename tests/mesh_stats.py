#!/usr/bin/env python3
"""Development aid: traversal counters of the mesh kernel (builds libsrt_pathtrace_dev_stats<N>.so: STATS=1 counters,
SRT_STATS_MODE=2 histogram by rays per phase, SRT_STATS_MODE=5 wave-cycles per phase and per wave life).
usage: python tests/mesh_stats.py [--mesh 224] [--spp 8] [--bounces 8] [--width 1920 --height 1080 --rows a,b] [--json]"""
import argparse, ctypes as C, importlib, json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--mesh", type=int, default=224)
ap.add_argument("--spp", type=int, default=8)
ap.add_argument("--bounces", type=int, default=8)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--rows", default="")
ap.add_argument("--json", action="store_true", help="print the counters as one JSON line")
a = ap.parse_args()
mode = int(os.environ.get("SRT_STATS_MODE", "1"))
srt = importlib.import_module("software-raytracer_amd")
srt.capi.use_dev_library(stats=mode)
L = srt.load_library()
path = os.path.join(ROOT, "software-raytracer_amd", "scenes", "Scene1.json")
sj = json.load(open(path))
sj["SceneObjects"][64]["Renderer"] = {"Type": "Mesh", "Primitive": "UVSphere", "Radius": 1.0, "Stacks": a.mesh, "Slices": a.mesh}
tmp = tempfile.NamedTemporaryFile("w", suffix=".json", delete=False); json.dump(sj, tmp); tmp.close()
sc = srt.host.Scene(tmp.name)
objs, n = sc.objects_copy(); meshes, nm = sc.meshes()
pt = srt.PathTracer(a.width, a.height)
pt.set_meshes(meshes, nm); pt.set_scene(objs, n); pt.set_camera(srt.default_camera())
rows = tuple(int(v) for v in a.rows.split(",")) if a.rows else None
out = (C.c_ulonglong * 8)()
for _ in range(2):  # the second launch uses the learned dispatch order; counters are read (and reset) after each
    pt.render(spp=a.spp, bounces=a.bounces, seed=0, count_rays=True, rows=rows)
    st = pt.stats()
    L.srt_debug_read_stats(out)
o = list(out)
if mode == 5:
    ph = max(o[6], 1)
    print("phases %d, %.1f rounds each, %.0f wave-cycles per phase = %.0f per round | in mesh phases: %.1f %% of all wave-cycles (%.3g of %.3g), kernel %.3f ms" %
          (o[6], o[4] / ph, o[0] / ph, o[0] / max(o[4], 1), 100.0 * o[0] / max(o[7], 1), o[0], o[7], st.kernel_ms))
    sys.exit(0)
if mode == 7:
    nr, lr = max(o[6], 1), max(o[7], 1)
    seg = [o[i] / nr for i in range(5)]
    print("node rounds %d: wave-cycles per round %.0f = pop %.0f + rows and ray %.0f + child tests %.0f + scan %.0f + pushes %.0f | leaf rounds %d: %.0f each | kernel %.3f ms" %
          (o[6], sum(seg), seg[0], seg[1], seg[2], seg[3], seg[4], o[7], o[5] / lr, st.kernel_ms))
    sys.exit(0)
if mode == 8:
    lr = max(o[7], 1)
    seg = [o[i] / lr for i in range(4)]
    print("leaf rounds %d: wave-cycles per round %.0f = pop %.0f + first rows and ray %.0f + first trip's tests and merge %.0f + later trips %.0f | kernel %.3f ms" %
          (o[7], sum(seg), seg[0], seg[1], seg[2], seg[3], st.kernel_ms))
    sys.exit(0)
if mode == 2:
    print("histogram by rays entering the phase (1-2, 3-8, 9-32, 33-64): phases", o[:4], "rounds", o[4:])
    sys.exit(0)
names = ["mesh phases (waves)", "go lanes", "node rounds", "node items", "leaf rounds", "leaf items", "overflows", "strict-mode entries"]
d = dict(zip(names, o))
d.update(rays=int(st.rays), kernel_ms=float(st.kernel_ms), path_samples=int(st.path_samples))
if a.json:
    print(json.dumps(d))
    sys.exit(0)
for k in names: print("%-22s %d" % (k, d[k]))
print("rays %d  kernel %.3f ms" % (st.rays, st.kernel_ms))
g = o[1] or 1
print("per mesh ray: node items %.1f leaf items %.1f (<= 4 triangles each, 4 lanes) | items per node round %.1f, leaves per leaf round %.1f | rounds per phase %.1f" %
      (o[3] / g, o[5] / g, o[3] / max(o[2], 1), o[5] / max(o[4], 1), (o[2] + o[4]) / max(o[0], 1)))
