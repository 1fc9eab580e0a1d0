#!/usr/bin/env python3
"""Prints DESIGN.md §4.7's results table from a round's committed files.  usage: python3 tools/design_table.py profiles/r04"""
import json, math, os, sys
d = sys.argv[1]
names = [("c2", "config 2: Scene1 1080p 32 spp 8 b (default)"), ("c3_rank0of8", "config 3, rank 0 of 8 equal bands: rows 0–135 (sky), 512 spp"), ("c3_rank4of8", "config 3, rank 4 of 8: rows 540–675"),
         ("c3_rank7of8", "config 3, rank 7 of 8: rows 945–1080 (floor)"), ("c3_rank4of8_probe", "config 3, rank 4 of 8 of the DEFAULT (balanced) split"), ("c4", "config 4: 99,904 triangles, 64 spp"),
         ("c5_rank4of8", "config 5, rank 4 of 8: 4K rows 1080–1350, 1024 spp, 16 b"), ("c5_rank5of8", "config 5, rank 5 of 8: rows 1350–1620 (the dearest band)"), ("scene_indirect", "Scene_indirect 1080p 32 spp 8 b")]
r3 = {"c2": 2.79e10, "c3_rank0of8": 6.08e10, "c3_rank4of8": 3.15e10, "c3_rank7of8": 1.61e10, "c3_rank4of8_probe": 1.20e10, "c4": 1.85e10, "c5_rank4of8": 2.18e10, "c5_rank5of8": 6.87e9, "scene_indirect": 5.06e9}
sup = {'0': '⁰', '1': '¹', '2': '²', '3': '³', '4': '⁴', '5': '⁵', '6': '⁶', '7': '⁷', '8': '⁸', '9': '⁹'}


def fmt(v):
    if not v:
        return "—"
    e = int(math.floor(math.log10(v)))
    return "%.2f·10%s" % (v / 10 ** e, "".join(sup[c] for c in str(e)))


print("| workload (`bench.py …`) | path-samples/s (round 3) | kernel ms, bench run (the run's first step / a fresh context's first launch, warm clocks) | traced run: mean of the same timed launches | launch shape | counted `roofline.frac` | VALU issue measured: by time at 2.4 GHz / by `GRBM_GUI_ACTIVE` cycles | model ÷ `SQ_INSTS_VALU` | HBM traffic ÷ algorithmic | CPU oracle, 16 threads |")
print("|---|---|---|---|---|---|---|---|---|---|")
vm = json.load(open(os.path.join(d, "valu_model.json")))
for n, label in names:
    b = json.loads(open(os.path.join(d, "bench_%s.json" % n)).read().strip().splitlines()[-1])
    rf = b["roofline"]
    dur = json.load(open(os.path.join(d, "kernel_durations_%s.json" % n)))["timed"]
    p = json.load(open(os.path.join(d, "pmc_%s.json" % n)))["timed"]
    sh = b["config"]["launch_shape"]
    shape = "one piece" if sh["grid_layers"] == 1 else "%d layers of %d" % (sh["grid_layers"], sh["chunk_samples"])
    cb = b.get("cpu_baseline", {})
    tr, ab = p.get("hbm_bytes", 0), b["roofline_hbm"]["algorithmic_bytes_per_launch"]
    g = lambda x: ("%.3g GB" % (x / 1e9)) if x > 1e9 else ("%.1f MB" % (x / 1e6))
    print("| %s | **%s** (%s) | %.2f (%.2f / %.2f) | %.2f | %s | %.3f | %.2f / %.2f | %.3f | %s / %s = %.2f | %s |" % (
        label, fmt(b["value"]), fmt(r3[n]), rf["kernel_ms"], rf["kernel_ms_cold_first_launch"], rf["kernel_ms_first_launch_warm_clocks"], dur["launch_ms_mean"], shape, rf["frac"],
        p["valu_issue_frac"], p["valu_issue_frac_grbm"], vm[n]["modelled_over_counted"], g(tr), g(ab), tr / ab, fmt(cb.get("value", 0))))
