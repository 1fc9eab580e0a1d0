#!/usr/bin/env python3
"""gpurun_out/<run>/emul_*.json (tools/emulate_ranks.py --json) -> profiles/emulated_ranks.json, the file bench.py quotes in the
JSON line of an N > 1 run.  usage: python3 tools/merge_emulated.py [--update] <emul_c3.json> <emul_c5.json> [held-out ones ...]
(--update keeps the entries of the existing file that are not replaced)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles", "emulated_ranks.json")
out = json.load(open(dst)) if "--update" in sys.argv and os.path.exists(dst) else {}
for path in [a for a in sys.argv[1:] if a != "--update"]:
    d = json.load(open(path))
    in_fit = d["config"] in (3, 5, "scene3")  # what round 4's joint fit of the balance weights used (tools/band_fit.py, profiles/r04/band_fit.txt)
    held_out = not in_fit
    key = "config %d" % d["config"] if isinstance(d["config"], int) else ("held out: " if held_out else "in the fit: ") + d["workload"]
    out.pop(key, None)
    e = out.setdefault(key, {"note": "every rank's launch run on its own, one after the other on ONE MI355X (tools/emulate_ranks.py): kernel ms per rank, "
                                     "no gather; whole frame on one GPU %.2f ms.  NOT a measured scaling curve.%s" %
                                     (d["whole_frame_ms"], "  HELD OUT: this workload was never used to fit the balance weights or the launch-shape rule." if held_out else "  Used in round 4's joint fit of the balance weights.")})
    for sp in d["splits"]:
        e.setdefault(str(sp["ranks"]), {})[sp["split"]] = {"bands": sp["bands"], "kernel_ms": [round(v, 3) for v in sp["kernel_ms"]],
                                                            "mean_over_slowest": round(sp["mean_over_slowest"], 3),
                                                            "whole_frame_over_n_over_slowest": round(sp["whole_over_n_over_slowest"], 3)}
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps({k: {n: {m: v[n][m]["mean_over_slowest"] for m in v[n]} for n in v if n != "note"} for k, v in out.items()}, indent=1))
