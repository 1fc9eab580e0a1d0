#!/usr/bin/env python3
"""rocprofv3 counters of one bench.py workload, one counter group per pass (GPU box only).

usage: python3 tools/pmc_collect.py <tag> [--groups hbm,sq1,sq2,sq3,sq4] -- <bench.py args>

Every pass is `rocprofv3 --pmc <group> -- python3 bench.py <args> --no-cpu-baseline` (counters never together
with a trace domain; FETCH_SIZE and WRITE_SIZE in passes of their own: together they exceed the TCC slots).
Only the FULL dispatches of the workload are summarised — those of the largest grid per kernel name; bench.py's
priming launch, the ray-count launch and parity re-renders on other grids are dropped.  Prints and writes
gpurun_out/<tag>/pmc.json: per kernel {dispatches, grid, mean of every counter} + derived figures:
  hbm_bytes            FETCH_SIZE x 1024 x 2 (gfx950: FETCH_SIZE reports half of a wide streaming read,
                       MI355X_MICROARCH.md §HBM) + WRITE_SIZE x 1024
  valu_issue_frac      SQ_INSTS_VALU x 2 cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)
  waves_per_cu         SQ_WAVE_CYCLES x 4 / (GRBM_GUI_ACTIVE / 8 x 256 CUs)   (occupancy actually held)
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

GROUPS = {
    "hbm_r": ["FETCH_SIZE"],
    "hbm_w": ["WRITE_SIZE"],
    "sq1": ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "GRBM_GUI_ACTIVE"],
    "sq2": ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"],
    "sq3": ["SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VMEM_WR", "SQ_WAIT_INST_LDS"],
    # the vector memory path (mesh kernels: gathers of BVH nodes and triangles)
    "tcp1": ["TCP_TOTAL_CACHE_ACCESSES", "TCP_TCC_READ_REQ", "TCP_TCC_READ_REQ_LATENCY", "TCP_PENDING_STALL_CYCLES"],
    "tcp2": ["TCP_TCP_LATENCY", "TCP_TOTAL_READ", "TCP_UTCL1_TRANSLATION_MISS", "TCP_UTCL1_TRANSLATION_HIT"],
    "tcc": ["TCC_HIT", "TCC_MISS", "TCC_REQ", "TCC_EA0_RDREQ"],
    "ta": ["TA_TA_BUSY", "TA_ADDR_STALLED_BY_TC_CYCLES", "TA_DATA_STALLED_BY_TC_CYCLES", "TA_TOTAL_WAVEFRONTS"],
}


def main():
    argv = sys.argv[1:]
    tag = argv[0]
    rest = argv[1:]
    groups = list(GROUPS)
    if rest and rest[0] == "--groups":
        groups = rest[1].split(",")
        rest = rest[2:]
    assert rest and rest[0] == "--", __doc__
    bench_args = rest[1:]
    root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = os.path.join(root, "gpurun_out", tag)
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    rows = collections.defaultdict(lambda: collections.defaultdict(list))  # kernel -> (grid, counter) -> values
    for g in groups:
        d = os.path.join(out, "pmc_" + g)
        cmd = ["rocprofv3", "--pmc"] + GROUPS[g] + ["--output-format", "csv", "-d", d, "--", "python3", os.path.join(root, "bench.py")] + bench_args + ["--no-cpu-baseline"]
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        open(os.path.join(out, "pmc_%s.log" % g), "w").write(r.stdout[-4000:] + "\n---- stderr ----\n" + r.stderr[-4000:])
        if r.returncode != 0:
            print("group %s failed (rc %d)" % (g, r.returncode))
            continue
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                name = row["Kernel_Name"]
                if "srt::" not in name:
                    continue
                short = name.split("(")[0].replace("void ", "")
                rows[short][(int(row.get("Grid_Size", 0) or 0), row["Counter_Name"])].append(float(row["Counter_Value"]))
    doc = {"bench_args": bench_args, "kernels": {}}
    for name, vals in rows.items():
        grid = max(g for g, _ in vals)  # only the workload's full-size dispatches
        k = {"grid_threads": grid}
        for (g, c), v in sorted(vals.items()):
            if g == grid:
                k[c] = sum(v) / len(v)
                k.setdefault("dispatches", len(v))
        if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
            k["hbm_bytes"] = k["FETCH_SIZE"] * 1024 * 2 + k["WRITE_SIZE"] * 1024
        if "GRBM_GUI_ACTIVE" in k and "SQ_INSTS_VALU" in k:
            k["valu_issue_frac"] = k["SQ_INSTS_VALU"] * 2 / (k["GRBM_GUI_ACTIVE"] / 8 * 1024)
        doc["kernels"][name] = k
    # counters of different passes: GRBM_GUI_ACTIVE is in sq1 only, use it for the sq2 ratios too
    for name, k in doc["kernels"].items():
        if "GRBM_GUI_ACTIVE" in k and "SQ_WAVE_CYCLES" in k:
            k["waves_per_cu"] = k["SQ_WAVE_CYCLES"] * 4 / (k["GRBM_GUI_ACTIVE"] / 8 * 256)
        if "SQ_WAVE_CYCLES" in k:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                if c in k:
                    k[c + "_per_wave_cycle"] = k[c] / k["SQ_WAVE_CYCLES"]
    json.dump(doc, open(os.path.join(out, "pmc.json"), "w"), indent=1)
    for name, k in doc["kernels"].items():
        print(name)
        for c, v in k.items():
            print("   %-36s %s" % (c, ("%.5g" % v) if isinstance(v, float) else v))


if __name__ == "__main__":
    main()
