#!/usr/bin/env python3
"""rocprofv3 counters of one bench.py workload, one counter group per pass (GPU box only).

usage: python3 tools/pmc_collect.py <tag> [--groups hbm_r,hbm_w,sq1,sq2] -- <bench.py args>

Every pass is `rocprofv3 --pmc <group> -- python3 bench.py <args> --no-cpu-baseline` (counters never together
with a trace domain; FETCH_SIZE and WRITE_SIZE in passes of their own: together they exceed the TCC slots).
Only the TIMED launches of the bench run are summarised, picked by POSITION (tools/dispatches.py: launch 0 is
bench.py's priming launch, then --warmup launches, then the --steps timed ones, whatever grid each uses; a launch =
one pathtrace_kernel dispatch + its fold_kernel when sample-chunked) — the launches the bench line times.  The warm-up
launches (the first is the frame's cold launch: other chunking, the dispatch-order probe ahead of it) are summarised
separately under "warmup".  Prints and writes gpurun_out/<tag>/pmc.json:
  timed.<counter>        mean per timed launch (pathtrace + fold), and per pass:
  passes.<group>.launch_ms_mean   device time of the SAME dispatches in the SAME pass (the csv's own timestamps) —
                         every ratio below uses the duration of the pass its counters came from
  hbm_bytes              FETCH_SIZE x 1024 x 2 (gfx950: FETCH_SIZE reports half of a wide streaming read,
                         MI355X_MICROARCH.md §HBM) + WRITE_SIZE x 1024
  valu_issue_frac        SQ_INSTS_VALU x 64 lanes / launch time of that pass / (256 CU x 128 lanes x 2.4 GHz)
  valu_issue_frac_grbm   SQ_INSTS_VALU x 2 cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)   (clock-independent)
  waves_per_cu           SQ_WAVE_CYCLES x 4 / (GRBM_GUI_ACTIVE / 8 x 256 CUs)   (occupancy actually held)
"""
import collections
import csv
import glob
import json
import os
import shutil
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


VALU_PEAK_LANEOPS = 256 * 128 * 2.4e9


def bench_arg(bench_args, flag, default):
    return int(bench_args[bench_args.index(flag) + 1]) if flag in bench_args else default


def main():
    argv = sys.argv[1:]
    tag = argv[0]
    rest = argv[1:]
    groups = ["hbm_r", "hbm_w", "sq1", "sq2"]
    if rest and rest[0] == "--groups":
        groups = rest[1].split(",")
        rest = rest[2:]
    assert rest and rest[0] == "--", __doc__
    bench_args = rest[1:]
    warmup, steps = bench_arg(bench_args, "--warmup", 2), bench_arg(bench_args, "--steps", 10)
    root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.join(root, "tools"))
    import dispatches as D

    out = os.path.join(root, "gpurun_out", tag)
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    doc = {"bench_args": bench_args, "warmup": warmup, "steps": steps, "selection": "by position: launch 0 priming, 1..warmup warm-up, then `steps` timed (tools/dispatches.py)",
           "passes": {}, "timed": {}, "warmup_launches": {}}
    for g in groups:
        d = os.path.join(out, "pmc_" + g)
        shutil.rmtree(d, ignore_errors=True)  # (csv files of an earlier run would be read as part of this pass)
        cmd = ["rocprofv3", "--pmc"] + GROUPS[g] + ["--output-format", "csv", "-d", d, "--", "python3", os.path.join(root, "bench.py")] + bench_args + ["--no-cpu-baseline"]
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        open(os.path.join(out, "pmc_%s.log" % g), "w").write(r.stdout[-4000:] + "\n---- stderr ----\n" + r.stderr[-4000:])
        if r.returncode != 0:
            print("group %s failed (rc %d)" % (g, r.returncode))
            continue
        launches = D.group_launches(D.read_counter_collection(glob.glob(d + "/**/*counter_collection.csv", recursive=True)))
        parts = D.classify(launches, warmup, steps)
        timed = parts["timed"]
        if len(timed) != steps:
            print("group %s: %d launches found, expected 1 + %d + %d + post" % (g, len(launches), warmup, steps))
            continue
        summ = D.summarise(timed)
        summ["launches_described"] = [D.describe(l) for l in timed[:2]]
        summ["warmup_launch_ms"] = [D.launch_ms(l) for l in parts["warmup"]]
        doc["passes"][g] = summ
        for c in GROUPS[g]:
            vals = [D.launch_counters(l).get(c) for l in timed]
            if all(v is not None for v in vals):
                doc["timed"][c] = sum(vals) / len(vals)
                doc["timed"][c + "__pass"] = g
            wv = [D.launch_counters(l).get(c) for l in parts["warmup"]]
            if wv and all(v is not None for v in wv):
                doc["warmup_launches"][c] = wv
    k, ps = doc["timed"], doc["passes"]
    if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
        k["hbm_bytes"] = k["FETCH_SIZE"] * 1024 * 2 + k["WRITE_SIZE"] * 1024
    if "SQ_INSTS_VALU" in k:
        ms = ps[k["SQ_INSTS_VALU__pass"]]["launch_ms_mean"]
        k["valu_issue_pass_launch_ms"] = ms
        k["valu_issue_frac"] = k["SQ_INSTS_VALU"] * 64 / (ms * 1e-3) / VALU_PEAK_LANEOPS
        if "GRBM_GUI_ACTIVE" in k:
            k["valu_issue_frac_grbm"] = k["SQ_INSTS_VALU"] * 2 / (k["GRBM_GUI_ACTIVE"] / 8 * 1024)
    # counters of different passes: GRBM_GUI_ACTIVE is in sq1 only, use it for the sq2 ratios too
    if "GRBM_GUI_ACTIVE" in k and "SQ_WAVE_CYCLES" in k:
        k["waves_per_cu"] = k["SQ_WAVE_CYCLES"] * 4 / (k["GRBM_GUI_ACTIVE"] / 8 * 256)
    if "SQ_WAVE_CYCLES" in k:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if c in k:
                k[c + "_per_wave_cycle"] = k[c] / k["SQ_WAVE_CYCLES"]
    if "SQ_INSTS_VALU" in k and "SQ_INSTS_SALU" in k:
        k["salu_per_valu"] = k["SQ_INSTS_SALU"] / k["SQ_INSTS_VALU"]
    json.dump(doc, open(os.path.join(out, "pmc.json"), "w"), indent=1, sort_keys=True)
    for g, p in ps.items():
        print("pass %-6s %d timed launches, %.3f ms each (%s)" % (g, p["launches"], p["launch_ms_mean"], ", ".join(p["kernels"])))
    for c, v in sorted(k.items()):
        print("   %-36s %s" % (c, ("%.5g" % v) if isinstance(v, float) else v))


if __name__ == "__main__":
    main()
