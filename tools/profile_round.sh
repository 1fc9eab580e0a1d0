#!/bin/bash
# Regenerates profiles/<round>/{kernel_stats.csv,pmc_summary.json} and profiles/traffic.json on the GPU box.
# usage (from the repo root, through gpurun):  bash tools/profile_round.sh r01
# Counters are collected in their own passes (no trace domains together with --pmc; FETCH_SIZE and
# WRITE_SIZE each alone — together they exceed the hardware's counter slots and rocprofv3 aborts).
# HBM bytes follow MI355X_MICROARCH.md, see the summarizer below.
set -e
ROUND=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profile_$ROUND
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/trace.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAVES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 120 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$tag -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_$tag.log 2>&1
done
python3 - "$ROUND" "$OUT" "$R" <<'PY'
import csv, glob, json, os, sys, collections
rnd, out, R = sys.argv[1:4]
os.makedirs(os.path.join(R, "profiles", rnd), exist_ok=True)
stats = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)
if stats:
    open(os.path.join(R, "profiles", rnd, "kernel_stats.csv"), "w").write(open(stats[0]).read())
vals = collections.defaultdict(list)
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "pathtrace_kernel" in row["Kernel_Name"]:
            vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
summ = {k: {"dispatches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for k, v in sorted(vals.items())}
doc = {"round": int(rnd.strip("r")), "kernel": "srt::pathtrace_kernel<4,false,true,false,false>", "workload": "Scene1 1920x1080 spp32 b8",
       "commands": ["rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline",
                    "rocprofv3 --pmc <one group per pass> --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"],
       "counters_per_dispatch": summ}
if "FETCH_SIZE" in summ and "WRITE_SIZE" in summ:
    # MI355X_MICROARCH.md: both counters are in KiB; on gfx950 FETCH_SIZE under-reports by 2x, WRITE_SIZE is exact
    fetch = summ["FETCH_SIZE"]["mean"] * 1024 * 2
    write = summ["WRITE_SIZE"]["mean"] * 1024
    doc["hbm_bytes_per_launch"] = {"fetch_corrected_x2": fetch, "write": write, "total": fetch + write}
    json.dump({"workload": doc["workload"], "hbm_bytes_per_launch": fetch + write, "fetch_bytes_corrected_x2": fetch, "write_bytes": write,
               "source": "profiles/%s/pmc_summary.json" % rnd}, open(os.path.join(R, "profiles", "traffic.json"), "w"), indent=1)
json.dump(doc, open(os.path.join(R, "profiles", rnd, "pmc_summary.json"), "w"), indent=1)
print(json.dumps({k: v["mean"] for k, v in summ.items()}, indent=1))
PY
cp $R/profiles/$ROUND/kernel_stats.csv $R/profiles/$ROUND/pmc_summary.json $R/profiles/traffic.json $OUT/ 2>/dev/null || true
