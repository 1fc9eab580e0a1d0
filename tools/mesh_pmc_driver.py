import ctypes as C, importlib, json, os, sys, tempfile
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
srt = importlib.import_module("software-raytracer_amd")
srt.capi.use_dev_library()
L = srt.load_library()
L.srt_debug_set_variant.argtypes = [C.c_void_p, C.c_int]
path = os.path.join(ROOT, "software-raytracer_amd", "scenes", "Scene1.json")
def run(edit, variant=-1, spp=16):
    sj = json.load(open(path)); edit(sj)
    tmp = tempfile.NamedTemporaryFile("w", suffix=".json", delete=False); json.dump(sj, tmp); tmp.close()
    sc = srt.host.Scene(tmp.name)
    objs, n = sc.objects_copy(); meshes, nm = sc.meshes()
    pt = srt.PathTracer(1920, 1080)
    pt.set_meshes(meshes, nm); pt.set_scene(objs, n); pt.set_camera(srt.default_camera())
    if variant >= 0: L.srt_debug_set_variant(pt._h, variant)
    pt.render(spp=spp, bounces=8, seed=0); pt.stats()
mesh = {"Type": "Mesh", "Primitive": "UVSphere", "Radius": 1.0, "Stacks": 224, "Slices": 224}
run(lambda sj: None)                                   # dispatch 1: analytic
def far(sj): sj["SceneObjects"].append({"Name": "m", "Position": [0, 0, -50], "Material": sj["SceneObjects"][64]["Material"], "Renderer": mesh})
run(far)                                               # 2: MESH kernel, no phases
def repl(sj): sj["SceneObjects"][64]["Renderer"] = mesh
run(repl, 101)                                         # 3: config 4, no deferral
run(repl, 116)                                         # 4: config 4, deferral 16
def small(sj): sj["SceneObjects"][64]["Renderer"] = dict(mesh, Stacks=16, Slices=16)
run(small, 116)                                        # 5: 480 tris
