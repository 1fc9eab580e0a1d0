#!/usr/bin/env python3
"""Development aid: kernel time of row bands against the forced sample-chunk size (development library, SRT_DEFER).
usage (GPU box): python3 tools/chunk_sweep.py <config 3|5> --bands 824:888,984:1080 [--chunks 0,256,128,64]
One child process per chunk size (the switch is read once per process); 0 = the library's own rule."""
import argparse, importlib, json, os, statistics, subprocess, sys, tempfile
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("config", type=int, choices=[2, 3, 4, 5, 6, 7, 8])
ap.add_argument("--bands", required=True)
ap.add_argument("--chunks", default="0,256,128,64")
ap.add_argument("--child", action="store_true")
a = ap.parse_args()
if not a.child:
    for c in a.chunks.split(","):
        env = dict(os.environ)
        if int(c):
            env["SRT_DEFER"] = c
        else:
            env.pop("SRT_DEFER", None)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), str(a.config), "--bands", a.bands, "--child"], env=env, capture_output=True, text=True, timeout=600)
        print("chunk %-4s %s" % (c if int(c) else "rule", r.stdout.strip() or r.stderr[-400:]), flush=True)
    sys.exit(0)
CFG = {2: ("Scene1", 0, 1920, 1080, 32, 8), 4: ("Scene1", 224, 1920, 1080, 64, 8), 8: ("Scene1", 0, 1920, 1080, 64, 8), 3: ("Scene1", 0, 1920, 1080, 512, 8), 5: ("Scene1", 224, 3840, 2160, 1024, 16), 6: ("Scene_indirect", 0, 1920, 1080, 512, 8), 7: ("Scene3", 0, 1920, 1080, 512, 8)}
scene, mesh, W, H, spp, bounces = CFG[a.config]
srt = importlib.import_module("software-raytracer_amd")
srt.capi.use_dev_library()
path = os.path.join(ROOT, "software-raytracer_amd", "scenes", scene + ".json")
if mesh:
    sj = json.load(open(path))
    sj["SceneObjects"][64]["Renderer"] = {"Type": "Mesh", "Primitive": "UVSphere", "Radius": 1.0, "Stacks": mesh, "Slices": mesh}
    tmp = tempfile.NamedTemporaryFile("w", suffix=".json", delete=False); json.dump(sj, tmp); tmp.close(); path = tmp.name
sc = srt.host.Scene(path)
if mesh:
    os.unlink(path)
objs, n = sc.objects_copy(); meshes, nm = sc.meshes()
out = []
for b in a.bands.split(","):
    rows = tuple(int(v) for v in b.split(":"))
    pt = srt.PathTracer(W, H)
    pt.set_meshes(meshes, nm); pt.set_scene(objs, n); pt.set_camera(srt.default_camera())
    ts = []
    for i in range(8):
        pt.render(spp=spp, bounces=bounces, seed=0, rows=rows)
        ts.append(pt.stats().kernel_ms)
    out.append("%s %.2f ms (%d chunks)" % (b, statistics.median(ts[3:]), pt.stats().sample_chunks))
    pt.close()
print(" | ".join(out))
