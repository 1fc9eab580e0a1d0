#!/usr/bin/env python3
"""Fits the weights of the multi-GPU balance probe (ProbeWeights in csrc/srt_capi.hip) in a closed loop.

  on the GPU box:   python3 tools/emulate_ranks.py <3|5> --modes probe --weights '<json>' --json out.json
        renders the bands of the 2 / 4 / 8-way splits that the given weights produce (one band after the other on one GPU) and
        stores, next to every band's kernel time, the probe's raw counts per block row (development library,
        srt_debug_probe_counts)
  anywhere:         python3 tools/band_fit.py [--fix=groups=660,waves=5830] out1.json out2.json ...
        least squares over all those runs: within every (run, N) group the bands should take the same time per unit of cost —
        log(cost_i / time_i) is to be the same for all bands i of a group — with the weights of the per-step part fixed
        (they only set the scale).  Prints the weights and each group's residuals; feed the weights to the next
        emulate_ranks run until the splits stop moving (round 3: four iterations, profiles/r03/band_fit_fit.txt).
"""
import json
import sys

import numpy as np
from scipy.optimize import least_squares

TALLY = ["steps", "groups", "node_rounds", "leaf_trips", "mesh_phases", "waves", "untraced_waves", "node_tests"]  # srt::TALLY_* (csrc/srt_kernel.hip.h), the first TALLY_N
FREE = ["groups", "node_rounds", "leaf_trips", "mesh_phases", "waves", "untraced_waves", "node_tests"]
PRIOR = dict(groups=300.0, node_rounds=26.0, leaf_trips=800.0, mesh_phases=65.0, waves=3000.0, untraced_waves=400.0, node_tests=30.0)  # where the search starts (a weak pull: 0.02)
STEP = dict(step=700.0, step_ugroup=70.0, step_cluster=12.0, step_box=45.0, step_mesh=60.0)  # read off the ISA, fixed


def step_weight(consts):
    ug, nc, nb, nt = consts
    return STEP["step"] + STEP["step_ugroup"] * ug + STEP["step_cluster"] * nc + STEP["step_box"] * nb + (STEP["step_mesh"] if nt else 0.0)


def band_counts(doc, rb, re):
    """count sums of memory rows [rb, re): block row j covers scene rows [16j, 16j+16) = memory rows (H-16j-16, H-16j]"""
    H = doc["height"]
    v = np.zeros(len(TALLY))
    for j, row in enumerate(doc["probe_rows"]):
        y0, y1 = 16 * j, min(16 * j + 16, H)
        m0, m1 = H - y1, H - y0
        ov = max(0, min(m1, re) - max(m0, rb))
        if ov:
            v += np.array(row, dtype=float) * (ov / float(y1 - y0))
    return v


def main(paths):
    fixed = {}
    if paths and paths[0].startswith("--fix="):  # --fix=groups=660,waves=5830: weights kept as given (e.g. fitted on another config)
        fixed = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in paths[0][len("--fix="):].split(",")}
        paths = paths[1:]
    free = [k for k in FREE if k not in fixed]
    groups = []
    for f in paths:
        d = json.load(open(f))
        sw = step_weight(d["consts"])
        for s in d["splits"]:
            if s["split"] != "probe":
                continue
            C = np.array([band_counts(d, *b) for b in s["bands"]])
            if C.shape[1] < len(TALLY):  # (round 3's dumps: seven counts and a spare word)
                C = np.pad(C, ((0, 0), (0, len(TALLY) - C.shape[1])))
            groups.append((f, s["ranks"], C, np.array(s["kernel_ms"]), sw))

    def cost(C, sw, wv):
        c = sw * C[:, 0]
        for k, v in list(zip(free, wv)) + list(fixed.items()):
            c = c + v * C[:, TALLY.index(k)]
        return c

    def resid(p):
        wv = np.exp(p)
        r = []
        for _, N, C, y, sw in groups:
            l = np.log(cost(C, sw, wv)) - np.log(y)
            r.append((l - l.mean()) * (1.0 if N > 2 else 0.5))
        return np.concatenate(r + [0.02 * (p - np.log([PRIOR[k] for k in free]))])  # weak pull towards the start

    sol = least_squares(resid, np.log([PRIOR[k] for k in free]), loss="soft_l1", f_scale=0.05)
    wv = np.exp(sol.x)
    print("weights: " + json.dumps({**{k: round(float(v), 1) for k, v in zip(free, wv)}, **fixed}))
    for f, N, C, y, sw in groups:
        l = np.log(cost(C, sw, wv)) - np.log(y)
        l -= l.mean()
        print("%s N=%d: cost/time of the bands against the group's mean, %%: %s   -> a split by this cost: mean / slowest about %.3f" %
              (f.split("/")[-1], N, " ".join("%+.1f" % (v * 100) for v in l), float(np.exp(l.min()))))


if __name__ == "__main__":
    main(sys.argv[1:])
