#!/usr/bin/env python3
"""Every rank's launch of an N-rank run, one after the other on ONE GPU, in one process (DESIGN.md §5's table).
usage (on the GPU box): python3 tools/emulate_ranks.py <config 3|5> [--ranks 2,4,8] [--modes probe,equal] [--json out.json]
For every N and split: each band is rendered by a context of its own (as a rank would: its own dispatch order, its own sample
chunking), 3 warm-up launches, then the median kernel time of 5.  Prints per-rank ms, mean / slowest, and the whole frame on one
GPU / N / slowest.  Kernel times only — no gather (<= 4.15 MB per rank)."""
import argparse, importlib, json, os, statistics, sys, tempfile
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("config", help="3 | 5 (the BASELINE configs the balance weights were fitted on) or a HELD-OUT workload: scene3 | scene_indirect | scene2 "
                                 "(1080p, 512 spp, 8 bounces) | c4_4k (config 4's mesh scene at 3840x2160, 256 spp, 8 bounces)")
ap.add_argument("--ranks", default="2,4,8")
ap.add_argument("--modes", default="probe,equal")
ap.add_argument("--json", default="")
ap.add_argument("--align", type=int, default=2, help="row alignment of the cost-balanced split")
ap.add_argument("--weights", default="", help="development: JSON of probe weights; the split is then computed here from the raw counts (development library)")
a = ap.parse_args()
CFG = {"3": ("Scene1", 0, 1920, 1080, 512, 8), "5": ("Scene1", 224, 3840, 2160, 1024, 16),
       # held out: never used for the fit of ProbeWeights or for the launch-shape rule (round 4, the verdict's item 5)
       "scene3": ("Scene3", 0, 1920, 1080, 512, 8), "scene_indirect": ("Scene_indirect", 0, 1920, 1080, 512, 8), "scene2": ("Scene2", 0, 1920, 1080, 512, 8),
       "c4_4k": ("Scene1", 224, 3840, 2160, 256, 8)}
scene, mesh, W, H, spp, bounces = CFG[a.config]
a.config = int(a.config) if a.config.isdigit() else a.config
label = ("config %d" % a.config) if isinstance(a.config, int) else a.config
srt = importlib.import_module("software-raytracer_amd")
stripes = importlib.import_module("software-raytracer_amd.stripes")
path = os.path.join(ROOT, "software-raytracer_amd", "scenes", scene + ".json")
if mesh:
    sj = json.load(open(path))
    sj["SceneObjects"][64]["Renderer"] = {"Type": "Mesh", "Primitive": "UVSphere", "Radius": 1.0, "Stacks": mesh, "Slices": mesh}
    tmp = tempfile.NamedTemporaryFile("w", suffix=".json", delete=False); json.dump(sj, tmp); tmp.close(); path = tmp.name
sc = srt.host.Scene(path)
if mesh:
    os.unlink(path)
objs, n = sc.objects_copy(); meshes, nm = sc.meshes()


def tracer():
    pt = srt.PathTracer(W, H)
    pt.set_meshes(meshes, nm); pt.set_scene(objs, n); pt.set_camera(srt.default_camera())
    return pt


def band_ms(rows):
    pt = tracer()
    ts = []
    for i in range(8):
        pt.render(spp=spp, bounces=bounces, seed=0, rows=rows)
        ts.append(pt.stats().kernel_ms)
    pt.close()
    return statistics.median(ts[3:])


TALLY = ["steps", "groups", "node_rounds", "leaf_trips", "mesh_phases", "waves", "untraced_waves", "node_tests"]  # srt::TALLY_* (the first TALLY_N)
KEYS = {"group": "groups", "node_round": "node_rounds", "leaf_trip": "leaf_trips", "mesh_phase": "mesh_phases", "wave": "waves", "untraced_wave": "untraced_waves", "node_test": "node_tests"}  # ProbeWeights' names
probe_rows = consts = None
if a.weights:
    import ctypes as C
    dev = srt.capi.open_library(os.path.join(ROOT, "software-raytracer_amd", "libsrt_pathtrace_dev.so"))
    dev.srt_debug_probe_counts.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    pd = srt.PathTracer(W, H, lib=dev)
    pd.set_meshes(meshes, nm); pd.set_scene(objs, n); pd.set_camera(srt.default_camera())
    nb = ((W + 15) // 16) * ((H + 15) // 16)
    NT = len(TALLY)
    buf = (C.c_uint32 * (NT * nb))(); bx, by = C.c_int(0), C.c_int(0); cc = (C.c_int * 4)()
    assert dev.srt_debug_probe_counts(pd._h, bounces, 0, buf, C.byref(bx), C.byref(by), cc) == 0
    pd.close()
    probe_rows = [[sum(buf[NT * (j * bx.value + i) + f] for i in range(bx.value)) for f in range(NT)] for j in range(by.value)]
    consts = list(cc)
    wt = json.loads(a.weights)
    stepw = wt.get("step", 700.0) + wt.get("step_ugroup", 70.0) * consts[0] + wt.get("step_cluster", 12.0) * consts[1] + wt.get("step_box", 45.0) * consts[2] + (wt.get("step_mesh", 60.0) if consts[3] else 0.0)
    row_cost = [0.0] * H
    for j, row in enumerate(probe_rows):
        c = stepw * row[0] + sum(v * row[TALLY.index(KEYS.get(k, k))] for k, v in wt.items() if KEYS.get(k, k) in TALLY)
        y0, y1 = 16 * j, min(16 * j + 16, H)
        for y in range(y0, y1):
            row_cost[H - 1 - y] = c / (y1 - y0)
else:
    pt = tracer()
    row_cost = pt.estimate_row_costs(bounces, 0)
    pt.close()
whole = band_ms((0, H))
print("%s: whole frame on one GPU %.2f ms" % (label, whole), flush=True)
doc = {"config": a.config, "workload": "%s%s %dx%d %d spp %d bounces" % (scene, "+mesh%d" % mesh if mesh else "", W, H, spp, bounces), "whole_frame_ms": whole, "splits": [], "probe_rows": probe_rows, "consts": consts, "height": H}
for N in (int(v) for v in a.ranks.split(",")):
    for mode in a.modes.split(","):
        bands = stripes.partition_rows(H, N, row_cost if mode == "probe" else None, align=a.align if mode == "probe" else 1)
        ms = [band_ms(b) for b in bands]
        eff = sum(ms) / N / max(ms)
        print("%s N=%d %-5s rows %s\n      ms %s | slowest %.2f  mean/slowest %.3f  (whole/N)/slowest %.3f" %
              (label, N, mode, [b[0] for b in bands] + [H], " / ".join("%.2f" % v for v in ms), max(ms), eff, whole / N / max(ms)), flush=True)
        doc["splits"].append({"ranks": N, "split": mode, "bands": [list(b) for b in bands], "kernel_ms": ms, "mean_over_slowest": eff, "whole_over_n_over_slowest": whole / N / max(ms)})
if a.json:
    json.dump(doc, open(a.json, "w"), indent=1)
