#!/usr/bin/env python3
"""gpurun_out/<run>/emul_c3.json + emul_c5.json (tools/emulate_ranks.py --json) -> profiles/emulated_ranks.json, the file bench.py
quotes in the JSON line of an N > 1 run.  usage: python3 tools/merge_emulated.py <emul_c3.json> <emul_c5.json> [more ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for path in sys.argv[1:]:
    d = json.load(open(path))
    key = "config %d" % d["config"]
    e = out.setdefault(key, {"note": "every rank's launch run on its own, one after the other on ONE MI355X (tools/emulate_ranks.py): kernel ms per rank, "
                                     "no gather; whole frame on one GPU %.2f ms.  NOT a measured scaling curve." % d["whole_frame_ms"]})
    for sp in d["splits"]:
        e.setdefault(str(sp["ranks"]), {})[sp["split"]] = {"bands": sp["bands"], "kernel_ms": [round(v, 3) for v in sp["kernel_ms"]],
                                                            "mean_over_slowest": round(sp["mean_over_slowest"], 3),
                                                            "whole_frame_over_n_over_slowest": round(sp["whole_over_n_over_slowest"], 3)}
json.dump(out, open(os.path.join(ROOT, "profiles", "emulated_ranks.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: {n: {m: v[n][m]["mean_over_slowest"] for m in v[n]} for n in v if n != "note"} for k, v in out.items()}, indent=1))
