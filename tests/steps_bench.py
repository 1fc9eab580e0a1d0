import importlib, os, sys, statistics
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
srt = importlib.import_module("software-raytracer_amd")
for scene in ("Scene1", "Scene_indirect"):
    sc = srt.host.Scene(os.path.join(ROOT, "software-raytracer_amd", "scenes", scene + ".json"))
    objs, n = sc.objects_copy()
    pt = srt.PathTracer(1920, 1080); pt.set_scene(objs, n); pt.set_camera(srt.default_camera())
    for preview in (False, True):
        out = []
        for steps in (1, 2, 4, 8):
            ts = []
            for i in range(8):
                pt.render(spp=1, bounces=8, seed=0, steps=steps, stripe_width=1920 // 16 + 1, preview=preview); ts.append(pt.stats().kernel_ms)
            out.append("steps %d: %.3f ms" % (steps, statistics.median(ts[2:])))
        print(scene, "preview" if preview else "path-traced 1 spp", " | ".join(out))
