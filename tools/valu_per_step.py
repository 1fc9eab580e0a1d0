#!/usr/bin/env python3
"""VALU instructions per pool step and per traced ray of every profiled workload, written INTO the round's pmc_<name>.json
(`derived`) and printed: SQ_INSTS_VALU per timed launch (the counter pass) over the run's own counts (the bench line's
`roofline.counted.pool_steps`; traced rays = rays - samples + pixels: the primary ray is traced once per pixel, every later ray once).
usage: python3 tools/valu_per_step.py profiles/r04        (tools/profile_round.py runs it at the end)

`history` (config 2 only): the same two figures at earlier points of the build, from the profiles committed then."""
import glob, json, os, re, sys

HISTORY_C2 = [  # (when, SQ_INSTS_VALU per timed launch) — same 1,261,717 pool steps and 74.42 M traced rays throughout
    ("round 3 (profiles/r03/pmc_c2.json)", 1.952e9),
    ("round 4, sphere tests' second halves per lane (commit a039e59)", 1.925e9),
    ("round 4, + split items, shifted-in 32-bit masks (commit of the first re-profile)", 1.769e9),
]


def derive(d):
    out = {}
    for bp in sorted(glob.glob(os.path.join(d, "bench_*.json"))):
        n = os.path.basename(bp)[6:-5]
        pp = os.path.join(d, "pmc_%s.json" % n)
        if not os.path.exists(pp):
            continue
        line = json.loads(open(bp).read().strip().splitlines()[-1])
        rf = line["roofline"]
        c = rf.get("counted") or {}
        doc = json.load(open(pp))
        valu = doc.get("timed", {}).get("SQ_INSTS_VALU")
        m = re.search(r"(\d+)x(\d+) rows(\d+)-(\d+)", line["config"].get("workload_key", ""))
        if not valu or not c.get("valid") or not m:
            continue
        samples = line["value"] * line["ms_per_step"] * 1e-3
        pixels = int(m.group(1)) * (int(m.group(4)) - int(m.group(3)))
        traced = rf["rays_per_sample"] * samples - samples + pixels
        dv = {"SQ_INSTS_VALU_per_timed_launch": valu, "pool_steps": c["pool_steps"], "traced_rays": round(traced),
              "valu_wave_instructions_per_pool_step": valu / c["pool_steps"], "valu_lane_instructions_per_traced_ray": valu * 64.0 / traced}
        if n == "c2":
            dv["history"] = [{"when": w, "SQ_INSTS_VALU_per_timed_launch": v, "valu_wave_instructions_per_pool_step": v / c["pool_steps"],
                              "valu_lane_instructions_per_traced_ray": v * 64.0 / traced} for w, v in HISTORY_C2]
        doc["derived"] = dv
        open(pp, "w").write(json.dumps(doc, indent=1, sort_keys=True))
        out[n] = dv
    return out


if __name__ == "__main__":
    for n, dv in derive(sys.argv[1]).items():
        print("%-20s %.4g VALU instructions per timed launch = %.0f per pool step = %.0f lane-instructions per traced ray" %
              (n, dv["SQ_INSTS_VALU_per_timed_launch"], dv["valu_wave_instructions_per_pool_step"], dv["valu_lane_instructions_per_traced_ray"]))
        for h in dv.get("history", []):
            print("    %-90s %.4g = %.0f per step = %.0f per ray" % (h["when"], h["SQ_INSTS_VALU_per_timed_launch"], h["valu_wave_instructions_per_pool_step"], h["valu_lane_instructions_per_traced_ray"]))
