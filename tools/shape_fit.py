#!/usr/bin/env python3
"""Development aid: read the launch-shape records a development library dumped (SRT_DUMP_RECORD, srt_capi.hip consume_record) and
replay the sample-chunk rule's inputs offline.

usage: python3 tools/shape_fit.py rec_c5.bin [rec_c3.bin ...] [--weights '{"leaf_trip": 900, ...}'] [--fit]

Per record (one band's recording launch): blocks, TIME figures (what round 3's rule read: wave time per block, the launch's event
time) and WORK figures (what the rule reads now: the waves' loop counts under the library's record weights): ratio = dearest block x slots / sum,
simulated fill for 1..8 layers.  --fit: least squares of the per-block wave time against the per-block counts (all records of
the given files together), to see what weights the blocks' times ask for."""
import json, struct, sys
import numpy as np

TALLY = ["step", "group", "node_round", "leaf_trip", "mesh_phase", "wave", "untraced_wave", "node_tests"]
# srt_capi.hip: k_record_weights_analytic / k_record_weights_mesh (what consume_record applies to the launch-shape record)
RECORD_WEIGHTS = {False: {"group": 110.0, "node_round": 0.0, "leaf_trip": 0.0, "mesh_phase": 0.0, "wave": 2500.0, "untraced_wave": 0.0, "node_tests": 0.0},
                  True: {"group": 127.0, "node_round": 0.0, "leaf_trip": 0.0, "mesh_phase": 33.0, "wave": 6900.0, "untraced_wave": 0.0, "node_tests": 34.0}}


def read(path):
    recs, data = [], open(path, "rb").read()
    off = 0
    while off < len(data):
        head = struct.unpack_from("<10I", data, off)
        assert head[0] == 0x53525452, "bad magic at %d" % off
        gx, gy, y0, rows = head[1:5]
        ms = struct.unpack("<f", struct.pack("<I", head[5]))[0]
        step_w = struct.unpack("<f", struct.pack("<I", head[6]))[0]
        n = gx * gy
        arr = np.frombuffer(data, dtype="<u4", count=n * 33, offset=off + 40)
        off += 40 + n * 33 * 4
        recs.append(dict(gx=gx, gy=gy, y0=y0, rows=rows, ms=ms, step_w=step_w, cu=head[7], mesh=bool(head[8]), has_work=bool(head[9]),
                         time=arr[:n].astype(np.float64), counts=arr[n:].reshape(n, 4, 8).astype(np.float64)))
    return recs


def simulate_fill(total, longest, layers, slots):
    import heapq
    order = np.argsort(-longest, kind="stable")
    heap = [0.0] * slots
    end = occ = 0.0
    for _ in range(layers):
        for i in order:
            t = heapq.heappop(heap) + longest[i] / layers
            end = max(end, t)
            heapq.heappush(heap, t)
            occ += total[i] / layers
    return occ / (end * slots * 4) if end > 0 else 1.0


def work(rec, w):
    wv = np.array([w.get("step", rec["step_w"])] + [w.get(k, RECORD_WEIGHTS[rec["mesh"]][k]) for k in TALLY[1:]])
    per_wave = rec["counts"] @ wv
    return per_wave.sum(axis=1), per_wave.max(axis=1)


def main():
    args = sys.argv[1:]
    w = {}
    if "--weights" in args:
        w = json.loads(args[args.index("--weights") + 1])
        del args[args.index("--weights"):args.index("--weights") + 2]
    fit = "--fit" in args
    if fit:
        args.remove("--fit")
    recs = [r for p in args for r in read(p)]
    for r in recs:
        n = r["gx"] * r["gy"]
        slots_ratio = r["cu"] * (3 if r["mesh"] else 4)
        slots = r["cu"] * (4 if r["mesh"] else 5)
        t = r["time"]
        tot, lng = work(r, w)
        tfill = t.sum() * 1e-5 / (r["ms"] * r["cu"] * (16 if r["mesh"] else 20)) if r["ms"] > 0 else 0
        line = "rows %4d-%4d %5d blocks %8.2f ms | TIME ratio %.3f fill %.3f | WORK ratio %.3f longest-ratio %.3f fill" % (
            r["y0"], r["y0"] + r["rows"], n, r["ms"], t.max() * slots_ratio / t.sum(), tfill, tot.max() * slots_ratio / tot.sum(), 4 * lng.max() * slots_ratio / tot.sum())
        for c in (1, 2, 4, 8):
            line += " (%d) %.3f" % (c, simulate_fill(tot, lng, c, slots))
        # how well the work predicts the per-block time (scale-free): correlation and the spread of time / work over the dear blocks
        dear = tot > 0.1 * tot.max()
        q = (t[dear] / tot[dear])
        line += " | corr %.3f  time/work of dear blocks p10 %.3g p50 %.3g p90 %.3g" % (np.corrcoef(t, tot)[0, 1], *np.percentile(q, [10, 50, 90]))
        print(line)
    if fit:
        X = np.concatenate([r["counts"].sum(axis=1) for r in recs])
        y = np.concatenate([r["time"] for r in recs])
        keep = X.sum(axis=1) > 0
        coef, *_ = np.linalg.lstsq(X[keep], y[keep], rcond=None)
        print("least squares, wave time (10 ns ticks) per count:", {k: round(float(v), 2) for k, v in zip(TALLY, coef)})
        print("  relative to a step = %.1f:" % recs[0]["step_w"], {k: round(float(v / coef[0] * recs[0]["step_w"]), 1) for k, v in zip(TALLY, coef)})


if __name__ == "__main__":
    main()
