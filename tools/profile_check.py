#!/usr/bin/env python3
"""Do a round's profile files describe the SAME launches?

For every workload <n> of profiles/<round>/ the kernel trace's summary (kernel_durations_<n>.json), every counter pass of
pmc_<n>.json and the committed bench line (bench_<n>.json) must name the same kernel instantiations for the TIMED launches, with
the same total grid (threads: x * y * z of the trace = Grid_Size of a counter pass) and the same number of grid layers as the
bench line's config.launch_shape.  Round 3's set failed exactly this for c5_rank4of8: the launch shape depended on run-time
timers, its trace ran the sample-chunked instantiation in 9 layers, its counter passes and its bench line the unchunked one.

usage: python3 tools/profile_check.py profiles/r04        (exit code 1 and one line per problem)
"""
import glob
import json
import os
import sys


def check_round(directory):
    problems = []
    for dur_path in sorted(glob.glob(os.path.join(directory, "kernel_durations_*.json"))):
        name = os.path.basename(dur_path)[len("kernel_durations_"):-len(".json")]
        dur = json.load(open(dur_path)).get("timed", {})
        kernels = sorted(dur.get("kernels", {}))
        grids = dur.get("grid_threads_total")
        if not kernels:
            problems.append("%s: the kernel trace has no timed launches" % name)
            continue
        pmc_path = os.path.join(directory, "pmc_%s.json" % name)
        if os.path.exists(pmc_path):
            for group, p in sorted(json.load(open(pmc_path)).get("passes", {}).items()):
                pk = sorted(p.get("kernels", {}))
                if pk != kernels:
                    problems.append("%s: counter pass %s ran %s, the kernel trace ran %s" % (name, group, pk, kernels))
                pg = p.get("grid_threads_total")
                if grids is not None and pg is not None and sorted(pg) != sorted(grids):
                    problems.append("%s: counter pass %s has grids of %s threads, the kernel trace %s" % (name, group, pg, grids))
        else:
            problems.append("%s: no pmc_%s.json next to the kernel trace" % (name, name))
        bench_path = os.path.join(directory, "bench_%s.json" % name)
        if os.path.exists(bench_path):
            line = json.loads(open(bench_path).read().strip().splitlines()[-1])
            shape = line.get("config", {}).get("launch_shape")
            if shape is None:
                problems.append("%s: the bench line has no config.launch_shape" % name)
            else:
                chunked = any("fold_kernel" in k for k in kernels)
                if chunked != (shape["grid_layers"] > 1):
                    problems.append("%s: the bench line ran %d grid layer(s), the kernel trace %s a fold kernel" % (name, shape["grid_layers"], "has" if chunked else "has not"))
                if not shape.get("same_as_timed_launches", True):
                    problems.append("%s: the bench line's counting launch had another shape than its timed launches" % name)
            frac = line.get("roofline", {}).get("frac")
            if frac is None or not (0 < frac <= 1.0):
                problems.append("%s: roofline.frac = %r is not a fraction in (0, 1]" % (name, frac))
        else:
            problems.append("%s: no bench_%s.json" % (name, name))
    return problems


if __name__ == "__main__":
    probs = check_round(sys.argv[1])
    for line in probs:
        print(line)
    sys.exit(1 if probs else 0)
