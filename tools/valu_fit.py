#!/usr/bin/env python3
"""Calibrates the three stated figures of bench.py's VALU_MODEL against the instruction counter, from a round's committed files.

usage: python3 tools/valu_fit.py profiles/r04

For every workload with a bench_<n>.json (its `roofline.counted` = the run's own loop counts) and a pmc_<n>.json
(SQ_INSTS_VALU per timed launch): wave-level VALU instructions = SQ_INSTS_VALU; priced tests = (24 x sphere tests + 35 x box tests +
14 x bound tests + 21 x BVH child tests + 40 x triangle tests + 30 x path-samples) / 64.  `step` = what config 2's counter leaves
per pool step; `bvh_round` and `mesh_phase` = least squares over the mesh workloads of what THEIR counters leave beyond step x
steps, per node round and per phase.  Prints the constants and every workload's modelled / counted ratio with them."""
import glob, json, os, sys
import numpy as np

M = {"sphere": 24, "box": 35, "triangle": 40, "bound": 14, "bvh_child": 21, "sample": 30}
d = sys.argv[1]
rows = {}
for bp in sorted(glob.glob(os.path.join(d, "bench_*.json"))):
    n = os.path.basename(bp)[6:-5]
    pp = os.path.join(d, "pmc_%s.json" % n)
    if not os.path.exists(pp):
        continue
    line = json.loads(open(bp).read().strip().splitlines()[-1])
    c = line["roofline"]["counted"]
    valu = json.load(open(pp))["timed"].get("SQ_INSTS_VALU")
    if not valu or not c.get("valid"):
        continue
    samples = line["value"] * line["ms_per_step"] * 1e-3
    tests = (M["sphere"] * (c["uniform_sphere_tests"] + c["cluster_sphere_tests"]) + M["box"] * c["box_tests"] + M["bound"] * c["cluster_bound_tests"] +
             M["bvh_child"] * c["bvh_child_tests"] + M["triangle"] * c["triangle_tests"] + M["sample"] * samples) / 64.0
    rows[n] = dict(valu=valu, tests=tests, steps=c["pool_steps"], rounds=c["bvh_node_rounds"], phases=c["mesh_phases"])
step = (rows["c2"]["valu"] - rows["c2"]["tests"]) / rows["c2"]["steps"]
mesh = [r for r in rows.values() if r["rounds"]]
bvh_round = mesh_phase = 0.0
if mesh:
    from scipy.optimize import nnls

    A = np.array([[r["rounds"], r["phases"]] for r in mesh], dtype=float)
    y = np.array([r["valu"] - r["tests"] - step * r["steps"] for r in mesh])
    sol, _ = nnls(A / y[:, None], np.ones(len(mesh)))  # non-negative, relative residuals
    bvh_round, mesh_phase = (float(v) for v in sol)
print("best fit: step = %.0f (config 2)   bvh_round = %.0f   mesh_phase = %.0f   (wave-level VALU instructions per pool step / node round / traversal phase beyond the priced tests)" % (step, bvh_round, mesh_phase))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

B = bench.VALU_MODEL
print("bench.py ships: step = %d   bvh_round = %d   mesh_phase = %d   (rounded DOWN to what the cheapest workload leaves: the model is a lower bound of the counter everywhere)" % (B["step"], B["bvh_round"], B["mesh_phase"]))
for n, r in rows.items():
    model = r["tests"] + step * r["steps"] + bvh_round * r["rounds"] + mesh_phase * r["phases"]
    shipped = r["tests"] + B["step"] * r["steps"] + B["bvh_round"] * r["rounds"] + B["mesh_phase"] * r["phases"]
    print("%-20s SQ_INSTS_VALU %.4g  best fit / counter %.3f   shipped model / counter %.3f   (priced tests %.0f %% of the counter; left per step %.0f)" %
          (n, r["valu"], model / r["valu"], shipped / r["valu"], 100 * r["tests"] / r["valu"], (r["valu"] - r["tests"]) / r["steps"]))
