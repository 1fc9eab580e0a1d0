#!/usr/bin/env python3
"""Development aid (GPU box): what launch shape srt_render chooses for a list of row bands, and the figures behind it.

usage: SRT_DEBUG_CHUNKS=1 python3 tools/shape_calibrate.py <config 3|4|5> rows[,rows...]   (rows = begin-end, memory rows)

Uses the development library: with SRT_DEBUG_CHUNKS set it prints, when a band's record arrives, round 3's TIME-based figures
(ratio of the dearest block's wave time to an even share, fill = wave time / launch time x resident waves — what the rule read
until round 3) next to the WORK-based ones the rule reads now (the same ratio over counted work, simulate_fill for 1..8 layers).
Per band: a fresh context, four launches; prints every launch's shape (srt_stats) and kernel time."""
import importlib, json, os, sys, tempfile
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("SRT_DEBUG_CHUNKS", "1")
cfg = int(sys.argv[1])
CFG = {3: ("Scene1", 0, 1920, 1080, 512, 8), 4: ("Scene1", 224, 1920, 1080, 64, 8), 5: ("Scene1", 224, 3840, 2160, 1024, 16),
       6: ("Scene_indirect", 0, 1920, 1080, 512, 8), 7: ("Scene3", 0, 1920, 1080, 512, 8), 9: ("Scene1", 224, 3840, 2160, 256, 8)}
scene, mesh, W, H, spp, bounces = CFG[cfg]
srt = importlib.import_module("software-raytracer_amd")
srt.capi.use_dev_library()
path = os.path.join(ROOT, "software-raytracer_amd", "scenes", scene + ".json")
if mesh:
    sj = json.load(open(path))
    sj["SceneObjects"][64]["Renderer"] = {"Type": "Mesh", "Primitive": "UVSphere", "Radius": 1.0, "Stacks": mesh, "Slices": mesh}
    tmp = tempfile.NamedTemporaryFile("w", suffix=".json", delete=False); json.dump(sj, tmp); tmp.close(); path = tmp.name
sc = srt.host.Scene(path)
objs, n = sc.objects_copy(); meshes, nm = sc.meshes()
for spec in sys.argv[2].split(","):
    rb, re = (int(v) for v in spec.split("-"))
    pt = srt.PathTracer(W, H)
    pt.set_meshes(meshes, nm); pt.set_scene(objs, n); pt.set_camera(srt.default_camera())
    print("== config %d rows %d-%d (%d spp, %d bounces)" % (cfg, rb, re, spp, bounces), flush=True)
    for i in range(int(os.environ.get('SRT_CAL_LAUNCHES', '4'))):
        pt.render(spp=spp, bounces=bounces, seed=0, rows=(rb, re))
        st = pt.stats()
        sys.stderr.flush()
        print("   launch %d: tile_rows %d layers %d chunk %d source %d  %.3f ms" % (i, st.tile_rows, st.sample_chunks, st.chunk_samples, st.shape_source, st.kernel_ms), flush=True)
    pt.close()
