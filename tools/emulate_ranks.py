#!/usr/bin/env python3
"""Every rank's launch of an N-rank run, one after the other on ONE GPU (DESIGN.md §5's table).
usage (on the GPU box): python3 tools/emulate_ranks.py <config> <N> <equal|probe> [--steps K]
Prints one line per rank (rows, kernel ms) and the slowest; kernel times only, no gather."""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg, n, mode = sys.argv[1], int(sys.argv[2]), sys.argv[3]
steps = sys.argv[5] if len(sys.argv) > 5 and sys.argv[4] == "--steps" else "5"
ms = []
for k in range(n):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--rank", "%d/%d" % (k, n), "--balance", mode, "--steps", steps, "--warmup", "2",
                        "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    js = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not js:
        print("rank %d failed: %s" % (k, r.stderr[-300:]), flush=True)
        sys.exit(1)
    d = json.loads(js[-1])
    ms.append(d["roofline"]["kernel_ms"])
    print("config %s N=%d %s rank %d rows %s: %.2f ms" % (cfg, n, mode, k, d["per_rank"][0]["rows"], ms[-1]), flush=True)
print("config %s N=%d %s: %s | slowest %.2f, mean %.2f, mean / slowest %.2f" % (cfg, n, mode, " / ".join("%.1f" % v for v in ms), max(ms), sum(ms) / n, sum(ms) / n / max(ms)), flush=True)
