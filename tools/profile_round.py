#!/usr/bin/env python3
"""Regenerates profiles/<round>/ and profiles/counters.json on the GPU box.

usage (through gpurun, from the repo root):  python3 tools/profile_round.py r04 [--only c2,c4,...]

For every workload below:
  profiles/<round>/bench_<name>.json         the bench.py line (roofline with its counted work, roofline_hbm, readback, cpu_baseline where it applies)
  profiles/<round>/kernel_stats_<name>.csv   rocprofv3 --kernel-trace --stats of the same bench.py command
  profiles/<round>/kernel_durations_<name>.json   the same trace launch by launch (tools/dispatches.py: timed launches by position)
  profiles/<round>/pmc_<name>.json           PMC counters per FULL launch of the workload, one counter group per pass
                                             (tools/pmc_collect.py: never together with a trace domain; FETCH_SIZE and
                                             WRITE_SIZE in passes of their own), summed over the launch's kernels
                                             (pathtrace_kernel + fold_kernel for sample-chunked launches); the TIMED launches are
                                             picked by position, every fraction is computed with the launch time of the pass the
                                             counter came from (stored next to it)
  profiles/counters.json[workload_key]       the same, keyed for bench.py: SQ_INSTS_VALU, hbm_bytes_per_launch = FETCH_SIZE x 1024 x 2
                                             + WRITE_SIZE x 1024 (gfx950: FETCH_SIZE reports half of a wide streaming read, MI355X_MICROARCH.md)
  profiles/<round>/valu_model.json           per workload: the VALU instructions rocprofv3 counted (SQ_INSTS_VALU per timed launch) next to
                                             what bench.py's VALU_MODEL makes of the run's own work counts — the check of the model
                                             behind roofline.frac
At the end tools/profile_check.py verifies that trace, counter passes and bench line of every workload describe the same launches
(kernel instantiation, grid, grid layers).  The launch shape is a function of the inputs (round 4), so they must.
"""
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import dispatches as D  # noqa: E402
WORKLOADS = [  # name, bench.py arguments, mesh tessellation
    ("c2", [], 0),
    ("c3_rank0of8", ["--config", "3", "--rank", "0/8", "--balance", "equal"], 0),
    ("c3_rank4of8", ["--config", "3", "--rank", "4/8", "--balance", "equal"], 0),
    ("c3_rank7of8", ["--config", "3", "--rank", "7/8", "--balance", "equal"], 0),
    ("c3_rank4of8_probe", ["--config", "3", "--rank", "4/8", "--balance", "probe"], 0),
    ("c4", ["--config", "4"], 224),
    ("c5_rank4of8", ["--config", "5", "--rank", "4/8", "--balance", "equal", "--steps", "3", "--warmup", "2"], 224),
    ("c5_rank5of8", ["--config", "5", "--rank", "5/8", "--balance", "equal", "--steps", "3", "--warmup", "2"], 224),
    ("scene_indirect", ["--scene", "Scene_indirect"], 0),
]


def run(cmd, **kw):
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, **kw)


def main():
    rnd = sys.argv[1]
    only = None
    if "--only" in sys.argv:
        only = set(sys.argv[sys.argv.index("--only") + 1].split(","))
    # --bench-only: keep the traces and counters of an earlier pass of this round, redo only the committed bench lines,
    # valu_model.json and the consistency check (after a change that touches bench.py's pricing but no kernel)
    bench_only = "--bench-only" in sys.argv
    # everything is written twice: into profiles/ (bench.py reads counters.json / mesh_counts.json from there) and into
    # gpurun_out/profiles/ — the only directory a gpurun call brings back; `cp -r gpurun_out/profiles/. profiles/` afterwards
    out = os.path.join(ROOT, "gpurun_out", "profiles", rnd)
    scratch = os.path.join(ROOT, "gpurun_out", "profile_" + rnd)
    os.makedirs(out, exist_ok=True)
    os.makedirs(scratch, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cpath = os.path.join(ROOT, "profiles", "counters.json")
    counters = json.load(open(cpath)) if os.path.exists(cpath) else {}
    vpath = os.path.join(ROOT, "profiles", rnd, "valu_model.json")
    valu_model = json.load(open(vpath)) if os.path.exists(vpath) else {}  # (a round is profiled in several gpurun calls: keep the others' entries)
    bench = [sys.executable, os.path.join(ROOT, "bench.py")]
    for name, args, mesh in WORKLOADS:
        if only and name not in only:
            continue
        steps = [] if "--steps" in args else ["--steps", "10", "--warmup", "3"]
        # 1. a first bench line: the workload key, rays per sample, samples per launch
        r = run(bench + args + steps + ["--no-cpu-baseline"], cwd=ROOT)
        if r.returncode != 0:
            print(name, "bench failed:", r.stderr[-500:])
            continue
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        key = line["config"]["workload_key"]
        print(name, key, "%.4g samples/s" % line["value"], flush=True)
        if bench_only:
            for kind in ("kernel_durations_%s.json", "pmc_%s.json", "kernel_stats_%s.csv"):  # (the check below reads them from `out`)
                src = os.path.join(ROOT, "profiles", rnd, kind % name)
                if os.path.exists(src) and not os.path.exists(os.path.join(out, kind % name)):
                    shutil.copy(src, os.path.join(out, kind % name))
        # 2. kernel trace + stats of the same command
        d = os.path.join(scratch, "trace_" + name)
        shutil.rmtree(d, ignore_errors=True)
        r = None if bench_only else run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", "python3", os.path.join(ROOT, "bench.py")] + args + steps +
                ["--no-cpu-baseline"], cwd="/tmp", env=env, timeout=900)
        st = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
        if st:
            open(os.path.join(out, "kernel_stats_%s.csv" % name), "w").write(open(st[0]).read())
        # the same trace, launch by launch (tools/dispatches.py): the TIMED launches are picked by position — after the priming
        # launch and the warm-up steps — whatever their grid, so a sample-chunked frame's steady state (fewer chunks than its
        # cold launches) is what gets summarised; priming / warm-up / post launches are listed apart
        tr = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
        if tr:
            n_warm, n_steps = D.bench_arg(args + steps, "--warmup", 2), D.bench_arg(args + steps, "--steps", 10)
            parts = D.classify(D.group_launches(D.read_kernel_trace(tr[0])), n_warm, n_steps)
            dur = {"selection": "by position: launch 0 priming, 1..%d warm-up, then %d timed, rest post (ray count, probes)" % (n_warm, n_steps),
                   "timed": D.summarise(parts["timed"]), "warmup": [D.describe(l) for l in parts["warmup"]],
                   "priming": [D.describe(l) for l in parts["priming"]], "post": [D.describe(l) for l in parts["post"]]}
            json.dump(dur, open(os.path.join(out, "kernel_durations_%s.json" % name), "w"), indent=1, sort_keys=True)
        # 3. counters, one group per pass; timed launches by position, every ratio with the duration of its own pass
        pmc_steps = ["--steps", "3", "--warmup", "3"]
        r = None if bench_only else run([sys.executable, os.path.join(ROOT, "tools", "pmc_collect.py"), "profile_%s/pmc_%s" % (rnd, name), "--groups", "hbm_r,hbm_w,sq1,sq2", "--"] +
                [a for i, a in enumerate(args) if a not in ("--steps", "--warmup") and (i == 0 or args[i - 1] not in ("--steps", "--warmup"))] + pmc_steps, cwd=ROOT, timeout=2400)
        pj = os.path.join(scratch, "pmc_" + name, "pmc.json")
        if os.path.exists(pj):
            doc = json.load(open(pj))
            k = doc.get("timed", {})
            entry = {"source": "profiles/%s/pmc_%s.json (tools/profile_round.py -> tools/pmc_collect.py: rocprofv3 --pmc, one group per pass, mean per TIMED "
                               "launch of bench.py picked by position; every fraction uses the launch time of the pass its counter came from)" % (rnd, name),
                     "kernels": sorted({kn for p in doc.get("passes", {}).values() for kn in p.get("kernels", {})}),
                     "pass_launch_ms": {g: p["launch_ms_mean"] for g, p in doc.get("passes", {}).items()}}
            for c, v in k.items():
                if not c.endswith("__pass"):
                    entry[c] = v
            if "hbm_bytes" in k:
                entry["hbm_bytes_per_launch"] = k["hbm_bytes"]
            counters[key] = entry
            json.dump(counters, open(cpath, "w"), indent=1, sort_keys=True)
            json.dump(counters, open(os.path.join(ROOT, "gpurun_out", "profiles", "counters.json"), "w"), indent=1, sort_keys=True)
            open(os.path.join(out, "pmc_%s.json" % name), "w").write(json.dumps(doc, indent=1, sort_keys=True))
        elif not bench_only:
            print(name, "pmc failed:", r.stdout[-300:], r.stderr[-300:])
        # 4. the bench line that is committed (reads the counters written above; with the CPU baseline where it is cheap)
        r = run(bench + args + steps, cwd=ROOT)
        js = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if js:
            open(os.path.join(out, "bench_%s.json" % name), "w").write(js[-1] + "\n")
            d = json.loads(js[-1])
            rf = d["roofline"]
            print("   -> %.4g samples/s, kernel %.3f ms (cold %.3f), counted valu frac %.3f, measured issue %s, hbm traffic %s vs algorithmic %d, shape %s" %
                  (d["value"] or 0, rf["kernel_ms"], rf["kernel_ms_cold_first_launch"] or 0, rf["frac"],
                   rf["measured_valu_issue_frac"], rf["traffic"], d["roofline_hbm"]["algorithmic_bytes_per_launch"], d["config"]["launch_shape"]), flush=True)
            # the model behind roofline.frac against the instruction counter: executed lane-ops / 64 = wave-level VALU instructions
            ctr = counters.get(key, {})
            if ctr.get("SQ_INSTS_VALU") and rf.get("executed_laneops_per_sample"):
                samples = d["value"] * d["ms_per_step"] * 1e-3
                modelled = rf["executed_laneops_per_sample"] * samples / 64.0
                valu_model[name] = {"workload_key": key, "SQ_INSTS_VALU_per_launch": ctr["SQ_INSTS_VALU"], "modelled_valu_instructions_per_launch": modelled,
                                    "modelled_over_counted": modelled / ctr["SQ_INSTS_VALU"], "pool_steps": rf["counted"]["pool_steps"],
                                    "valu_model": rf["valu_model"], "counted": rf["counted"],
                                    "unpriced_instructions_per_step": (ctr["SQ_INSTS_VALU"] - (modelled - (rf["valu_model"]["step"] * rf["counted"]["pool_steps"]))) / rf["counted"]["pool_steps"]}
                json.dump(valu_model, open(os.path.join(out, "valu_model.json"), "w"), indent=1, sort_keys=True)
                print("      VALU instructions: counter %.4g, model %.4g (%.3f); what the tests leave per step: %.0f" %
                      (ctr["SQ_INSTS_VALU"], modelled, modelled / ctr["SQ_INSTS_VALU"], valu_model[name]["unpriced_instructions_per_step"]), flush=True)
        else:
            print(name, "final bench failed:", r.stderr[-500:])
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import profile_check as PC
    import valu_per_step  # VALU instructions per pool step / per traced ray into every pmc_<name>.json (`derived`)
    valu_per_step.derive(out)
    probs = PC.check_round(out)
    print("profile_check: %s" % ("all workloads describe the same launches" if not probs else "\n  ".join(["PROBLEMS"] + probs)), flush=True)


if __name__ == "__main__":
    main()
