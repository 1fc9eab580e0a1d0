#!/usr/bin/env python3
"""Fits the weights of the multi-GPU balance cost (srt_estimate_row_costs / BAL_W_* in csrc/srt_kernel.hip.h).

  on the GPU box:   python3 tools/band_fit.py measure gpurun_out/band_fit.json
        for every workload: the probe's raw per-block-row feature sums (development library, srt_debug_block_features) and
        the measured steady-state kernel time of many row bands (the bands an N = 2 / 4 / 8 run would launch, equal split and
        shifted by half a band), each band launched on its own like a rank would
  anywhere:         python3 tools/band_fit.py fit gpurun_out/band_fit.json
        least squares of  band_ms ~ scale_workload x sum over the band's rows of (w . features)  over all workloads, prints
        the weights scaled to BAL_W_RAY = 64 and, for N = 2 / 4 / 8, the emulated mean / slowest of the split the fitted
        weights produce (band times interpolated from the measured per-row cost density).
"""
import ctypes as C
import importlib
import json
import os
import statistics
import sys
import tempfile

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
WORK = {  # name: scene, mesh, W, H, spp, bounces
    "c3": ("Scene1", 0, 1920, 1080, 512, 8),
    "c5": ("Scene1", 224, 3840, 2160, 1024, 16),
    "indirect": ("Scene_indirect", 0, 1920, 1080, 512, 8),
    "scene3": ("Scene3", 0, 1920, 1080, 512, 8),
    "c4x8": ("Scene1", 224, 1920, 1080, 512, 8),
}
FEATURES = ["pixels", "traced", "rays", "cand", "mesh_go", "miss", "mesh_hit"]


def scene_file(scene, mesh):
    path = os.path.join(ROOT, "software-raytracer_amd", "scenes", scene + ".json")
    if not mesh:
        return path, False
    sj = json.load(open(path))
    sj["SceneObjects"][64]["Renderer"] = {"Type": "Mesh", "Primitive": "UVSphere", "Radius": 1.0, "Stacks": mesh, "Slices": mesh}
    tmp = tempfile.NamedTemporaryFile("w", suffix=".json", delete=False)
    json.dump(sj, tmp)
    tmp.close()
    return tmp.name, True


def measure(out_path, names):
    srt = importlib.import_module("software-raytracer_amd")
    prod = srt.capi.open_library(os.path.join(ROOT, "software-raytracer_amd", "libsrt_pathtrace.so"))
    dev = srt.capi.open_library(os.path.join(ROOT, "software-raytracer_amd", "libsrt_pathtrace_dev.so"))
    dev.srt_debug_block_features.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    doc = {}
    for name in names:
        scene, mesh, W, H, spp, bounces = WORK[name]
        path, tmp = scene_file(scene, mesh)
        sc = srt.host.Scene(path)
        if tmp:
            os.unlink(path)
        objs, n = sc.objects_copy()
        meshes, nm = sc.meshes()

        def tracer(lib):
            pt = srt.PathTracer(W, H, lib=lib)
            pt.set_meshes(meshes, nm)
            pt.set_scene(objs, n)
            pt.set_camera(srt.default_camera())
            return pt

        pd = tracer(dev)
        nb = ((W + 15) // 16) * ((H + 15) // 16)
        buf = (C.c_uint32 * (8 * nb))()
        bx, by = C.c_int(0), C.c_int(0)
        rc = dev.srt_debug_block_features(pd._h, bounces, 0, buf, C.byref(bx), C.byref(by))
        assert rc == 0, rc
        # per block row j (scene rows [16 j, 16 j + 16) -> memory rows H - 16 j - 16 .. H - 16 j): feature sums
        rows = []
        for j in range(by.value):
            rows.append([sum(buf[8 * (j * bx.value + i) + f] for i in range(bx.value)) for f in range(7)])
        blocks = [[int(buf[8 * b + f]) for f in range(7)] for b in range(bx.value * by.value)]  # per 16 x 16 block, row-major from scene row 0
        pd.close()
        pt = tracer(prod)
        bands = []
        for N in (8, 4, 2):
            h = H // N
            cuts = [(k * h, (k + 1) * h) for k in range(N)]
            if N > 2:
                cuts += [(k * h + h // 2, (k + 1) * h + h // 2) for k in range(N - 1)]
            for rb, re in cuts:
                ts = []
                for i in range(6):
                    pt.render(spp=spp, bounces=bounces, seed=0, rows=(rb, re))
                    ts.append(pt.stats().kernel_ms)
                bands.append({"rows": [rb, re], "ms": statistics.median(ts[3:]), "ms_all": ts, "sample_chunks": int(pt.stats().sample_chunks)})
                print(name, "rows", rb, re, "%.3f ms" % bands[-1]["ms"], flush=True)
        pt.close()
        doc[name] = {"scene": scene, "mesh": mesh, "width": W, "height": H, "spp": spp, "bounces": bounces, "block_rows": rows, "blocks": blocks, "blocks_x": bx.value, "bands": bands}
        json.dump(doc, open(out_path, "w"))


def band_features(w, rb, re):
    """feature sums of memory rows [rb, re): block row j covers scene rows [16j, 16j+16) = memory rows (H-16j-16, H-16j]"""
    import numpy as np

    H = w["height"]
    f = np.zeros(7)
    for j, row in enumerate(w["block_rows"]):
        y0, y1 = 16 * j, min(16 * j + 16, H)
        m0, m1 = H - y1, H - y0  # memory rows [m0, m1)
        ov = max(0, min(m1, re) - max(m0, rb))
        if ov:
            f += np.array(row, dtype=float) * (ov / float(y1 - y0))
    return f


def fit(path, use):
    """joint fit over the workloads: log(band_ms) ~ log(scale_workload) + log(sum over rows of w . features), w_rays = 64"""
    import numpy as np
    from scipy.optimize import least_squares

    doc = json.load(open(path))
    names = [n for n in doc if not use or n in use]
    cols = ["pixels", "rays", "cand", "mesh_go", "mesh_hit"]
    idx = [FEATURES.index(c) for c in cols]
    X = {n: np.array([band_features(doc[n], *b["rows"])[idx] for b in doc[n]["bands"]]) for n in names}
    y = {n: np.array([b["ms"] for b in doc[n]["bands"]]) for n in names}
    free = [i for i, c in enumerate(cols) if c != "rays" and any(X[n][:, i].any() for n in names)]

    def unpack(p):
        w = np.zeros(len(cols))
        w[1] = 64.0
        w[free] = np.exp(p[:len(free)])
        return w, dict(zip(names, p[len(free):]))

    def resid(p):
        w, ls = unpack(p)
        return np.concatenate([ls[n] + np.log(X[n] @ w) - np.log(y[n]) for n in names])

    p0 = np.concatenate([np.log([7.0, 8.0, 100.0, 200.0])[:len(free)], [np.log(y[n].sum() / (X[n][:, 1].sum() * 64.0)) for n in names]])
    sol = least_squares(resid, p0, loss="soft_l1", f_scale=0.1)
    w, ls = unpack(sol.x)
    print("weights (BAL_W_RAY = 64): " + ", ".join("%s %.1f" % (c, v) for c, v in zip(cols, w)))
    for n in names:
        err = np.exp(ls[n] + np.log(X[n] @ w) - np.log(y[n])) - 1
        print("%-9s scale %.3e  rel. error of the band times: rms %.3f  max %.3f" % (n, float(np.exp(ls[n])), float(np.sqrt((err ** 2).mean())), float(np.abs(err).max())))
    return w, cols


def emulate(path, w_by_col, use):
    """for N in 2/4/8: split by the weighted row cost; band time = integral of the MEASURED per-row time density (from the
    finest measured bands, N = 8 equal + shifted) over the band -> mean / slowest"""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "software-raytracer_amd"))
    stripes = importlib.import_module("software-raytracer_amd.stripes")
    doc = json.load(open(path))
    for n, wl in doc.items():
        if use and n not in use:
            continue
        H = wl["height"]
        h8 = H // 8
        dens = np.zeros(H)
        cnt = np.zeros(H)
        for b in wl["bands"]:
            rb, re = b["rows"]
            if re - rb == h8:
                dens[rb:re] += b["ms"] / (re - rb)
                cnt[rb:re] += 1
        dens /= np.maximum(cnt, 1)
        idx = [FEATURES.index(c) for c in w_by_col]
        wv = np.array([w_by_col[c] for c in w_by_col])
        row_cost = [float(band_features(wl, m, m + 1)[idx] @ wv) for m in range(H)]
        for N in (2, 4, 8):
            bands = stripes.partition_rows(H, N, row_cost, align=8)
            t = [float(dens[a:b].sum()) for a, b in bands]
            eq = [float(dens[k * (H // N):(k + 1) * (H // N)].sum()) for k in range(N)]
            print("%-9s N=%d  weighted split: mean/slowest %.3f   equal bands: %.3f   (density model; bands %s)" %
                  (n, N, sum(t) / N / max(t), sum(eq) / N / max(eq), bands if N == 8 else ""))


if __name__ == "__main__":
    mode, path = sys.argv[1], sys.argv[2]
    rest = sys.argv[3:]
    if mode == "measure":
        measure(path, rest or ["c3", "c5", "indirect", "scene3", "c4x8"])
    else:
        w, cols = fit(path, rest)
        emulate(path, dict(zip(cols, w)), rest)
