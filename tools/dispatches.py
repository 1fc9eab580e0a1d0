"""Which kernel dispatches of a rocprofv3 run of bench.py belong to which of bench.py's launches.

bench.py (N = 1) makes its srt_render calls in a fixed order:

    1 priming launch (8 rows, 1 spp)  |  W warm-up steps  |  K TIMED steps  |  1 ray-count launch
    [| 1 one-sample probe when the launch is sample-chunked]  [| parity re-renders of the cpu_baseline leg]

A "launch" here is one srt_render call as the device sees it: ONE dispatch of some srt::pathtrace_kernel<...>
instantiation, the srt::fold_kernel dispatch that follows it when the launch is sample-chunked, and — ahead of a
frame's first launch only — the dispatch-order probe (block_cost / smooth_cost / order_sort kernels), kept apart
as `aux`.  Launches are numbered in dispatch order, so launch 0 is the priming launch, 1..W the warm-up (the first
of them cold: other chunking, other grid z), W+1..W+K the timed steps — whatever grid each of them uses.  That is
the selection the profiles need: the steady-state launches the bench line times, not "the largest grid" (a
sample-chunked frame's first launches have MORE chunks than its steady state and used to be picked instead).
"""
import csv


def short_name(kernel_name):
    return kernel_name.split("(")[0].replace("void ", "").strip()


def read_kernel_trace(path):
    """rocprofv3 --kernel-trace csv -> list of dispatches {id, kernel, start, end, grid:(x,y,z) in threads}"""
    out = []
    for row in csv.DictReader(open(path)):
        if "srt::" not in row["Kernel_Name"]:
            continue
        out.append({"id": int(row["Dispatch_Id"]), "kernel": short_name(row["Kernel_Name"]), "start": int(row["Start_Timestamp"]),
                    "end": int(row["End_Timestamp"]),
                    "grid": tuple(int(row.get("Grid_Size_" + ax, 1) or 1) for ax in "XYZ"), "counters": {}})
    return sorted(out, key=lambda d: d["id"])


def read_counter_collection(paths):
    """rocprofv3 --pmc csv files of ONE pass -> list of dispatches with their counters (one entry per dispatch)"""
    by_id = {}
    for path in paths:
        for row in csv.DictReader(open(path)):
            if "srt::" not in row["Kernel_Name"]:
                continue
            key = (int(row.get("Process_Id", 0) or 0), int(row["Dispatch_Id"]))
            d = by_id.setdefault(key, {"id": key[1], "pid": key[0], "kernel": short_name(row["Kernel_Name"]), "start": int(row.get("Start_Timestamp", 0) or 0),
                                       "end": int(row.get("End_Timestamp", 0) or 0), "grid": (int(row.get("Grid_Size", 0) or 0), 1, 1), "counters": {}})
            # (a counter may be reported in several rows — one per instance dimension; they add up)
            d["counters"][row["Counter_Name"]] = d["counters"].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    # a directory that was written to by several runs holds several processes: only the LAST run counts
    last = max((d["start"], d["pid"]) for d in by_id.values())[1] if by_id else 0
    return sorted((d for d in by_id.values() if d["pid"] == last), key=lambda d: d["id"])


def bench_arg(bench_args, flag, default):
    return int(bench_args[bench_args.index(flag) + 1]) if flag in bench_args else default


def group_launches(dispatches):
    """dispatches in order -> launches [{main, fold, aux, kernels:[...]}] as described in the module docstring"""
    launches, aux = [], []
    for d in dispatches:
        if "pathtrace_kernel" in d["kernel"]:
            launches.append({"main": d, "fold": None, "aux": aux})
            aux = []
        elif "fold_kernel" in d["kernel"] and launches and launches[-1]["fold"] is None:
            launches[-1]["fold"] = d
        else:
            aux.append(d)
    return launches


def classify(launches, warmup, steps):
    """-> {"priming": [...], "warmup": [...], "timed": [...], "post": [...]} by position"""
    return {"priming": launches[:1], "warmup": launches[1:1 + warmup], "timed": launches[1 + warmup:1 + warmup + steps], "post": launches[1 + warmup + steps:]}


def launch_ms(launch):
    """device time of the launch's own kernels (pathtrace + fold), in ms"""
    ms = (launch["main"]["end"] - launch["main"]["start"]) * 1e-6
    if launch["fold"]:
        ms += (launch["fold"]["end"] - launch["fold"]["start"]) * 1e-6
    return ms


def launch_counters(launch):
    tot = dict(launch["main"]["counters"])
    if launch["fold"]:
        for c, v in launch["fold"]["counters"].items():
            tot[c] = tot.get(c, 0.0) + v
    return tot


def describe(launch):
    d = {"kernel": launch["main"]["kernel"], "dispatch_id": launch["main"]["id"], "grid_threads": list(launch["main"]["grid"]),
         "ms": (launch["main"]["end"] - launch["main"]["start"]) * 1e-6}
    if launch["fold"]:
        d["fold_dispatch_id"] = launch["fold"]["id"]
        d["fold_ms"] = (launch["fold"]["end"] - launch["fold"]["start"]) * 1e-6
    if launch["aux"]:
        d["aux"] = [{"kernel": a["kernel"], "ms": (a["end"] - a["start"]) * 1e-6} for a in launch["aux"]]
    return d


def summarise(launches):
    """mean / min / max device ms of a list of launches, per kernel and per launch"""
    if not launches:
        return {"launches": 0}
    per = {}
    for l in launches:
        per.setdefault(l["main"]["kernel"], []).append((l["main"]["end"] - l["main"]["start"]) * 1e-6)
        if l["fold"]:
            per.setdefault(l["fold"]["kernel"], []).append((l["fold"]["end"] - l["fold"]["start"]) * 1e-6)
    tot = [launch_ms(l) for l in launches]
    def threads(g):
        n = 1
        for v in g:
            n *= max(int(v), 1)
        return n

    return {"launches": len(launches), "launch_ms_mean": sum(tot) / len(tot), "launch_ms_min": min(tot), "launch_ms_max": max(tot),
            "dispatch_ids": [l["main"]["id"] for l in launches],
            # total threads of the launches' pathtrace dispatches (x * y * z: the kernel trace reports the three sizes, a counter pass
            # their product) — with the kernel names, what tools/profile_check.py compares between a trace and a counter pass
            "grid_threads_total": sorted({threads(l["main"]["grid"]) for l in launches}),
            "kernels": {k: {"calls": len(v), "mean_ms": sum(v) / len(v), "min_ms": min(v), "max_ms": max(v)} for k, v in sorted(per.items())}}
