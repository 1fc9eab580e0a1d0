#!/usr/bin/env python3
"""The block-cost timer anomaly: records of ~2^32 in wg_cost, i.e. a wave whose end-minus-start s_memtime
difference is negative.  The development library built with STATS=6 logs for EVERY wave of a launch: s_memtime at start and
end (what __builtin_readcyclecounter lowers to on gfx950), s_memrealtime (the constant 100 MHz clock) at both points, and the
HW_ID / XCC_ID registers at both points (where the wave ran).  This script renders a few launches and reports:
  * waves with a non-positive or absurd s_memtime difference, with where they ran and their realtime difference,
  * whether such waves changed XCC / CU between start and end (wave save/restore would show up here),
  * per-XCC offsets of the s_memtime counter (min start per XCC relative to the realtime clock).
usage (GPU box): python3 tools/timer_probe.py [--launches 6] [--work c2|c3r7]     (also run it under `rocprofv3 --pmc SQ_WAVES --`)
"""
import argparse
import collections
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--launches", type=int, default=6)
ap.add_argument("--work", default="c2")
a = ap.parse_args()
srt = importlib.import_module("software-raytracer_amd")
L = srt.capi.open_library(os.path.join(ROOT, "software-raytracer_amd", "libsrt_pathtrace_dev_stats6.so"))
L.srt_debug_read_wave_log.argtypes = [C.c_void_p, C.c_size_t]
W, H = 1920, 1080
spp, rows = (32, (0, 1080)) if a.work == "c2" else (512, (945, 1080))
sc = srt.host.Scene(os.path.join(ROOT, "software-raytracer_amd", "scenes", "Scene1.json"))
objs, n = sc.objects_copy()
pt = srt.PathTracer(W, H, lib=L)
pt.set_scene(objs, n)
pt.set_camera(srt.default_camera())
MAXW = 1 << 17
buf = np.zeros((MAXW, 6), dtype=np.uint64)


def hw_decode(v):
    hw, xcc = int(v) & 0xFFFFFFFF, int(v) >> 32
    # gfx9 HW_ID: wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx90a+: [14:13]), ...
    return {"wave": hw & 15, "simd": (hw >> 4) & 3, "cu": (hw >> 8) & 15, "sh": (hw >> 12) & 1, "se": (hw >> 13) & 7, "xcc": xcc & 15, "raw": "%08x" % hw}


bad_total = 0
for it in range(a.launches):
    buf[:] = 0
    pt.render(spp=spp, bounces=8, seed=0, rows=rows)
    st = pt.stats()
    assert L.srt_debug_read_wave_log(buf.ctypes.data_as(C.c_void_p), MAXW) == 0
    live = buf[:, 1] != 0
    t0, t1, r0, r1, h0, h1 = (buf[live, i] for i in range(6))
    dt = t1.astype(np.int64) - t0.astype(np.int64)
    dr = r1.astype(np.int64) - r0.astype(np.int64)
    bad = (dt <= 0) | (dt > (1 << 40))
    moved = h0 != h1
    xcc = (h0 >> np.uint64(32)).astype(np.int64) & 15
    print("launch %d: %.3f ms, %d waves logged, s_memtime dt: median %d max %d min %d | realtime dt (10 ns ticks): median %d max %d | anomalous %d | HW_ID/XCC changed %d" %
          (it, st.kernel_ms, int(live.sum()), int(np.median(dt)), int(dt.max()), int(dt.min()), int(np.median(dr)), int(dr.max()), int(bad.sum()), int(moved.sum())))
    # the s_memtime counter of every XCC against the realtime clock: offset = t_start - 21 * r_start (about 2.1 GHz vs 100 MHz) is
    # only indicative; what matters is whether XCCs differ by more than a launch's length
    per = collections.OrderedDict()
    for x in sorted(set(xcc.tolist())):
        m = xcc == x
        per[x] = (int(t0[m].min()), int(r0[m].min()), int(m.sum()))
    base_t, base_r = per[next(iter(per))][:2]
    print("   per XCC (waves, first s_memtime - XCC0's, first realtime - XCC0's): " + "  ".join("%d:(%d, %+d, %+d)" % (x, c, t - base_t, r - base_r) for x, (t, r, c) in per.items()))
    for i in np.nonzero(bad)[0][:12]:
        print("   anomaly: dt %d  realtime dt %d  start %s  end %s" % (int(dt[i]), int(dr[i]), hw_decode(h0[i]), hw_decode(h1[i])))
    for i in np.nonzero(moved & ~bad)[0][:4]:
        print("   moved (dt fine): dt %d  start %s  end %s" % (int(dt[i]), hw_decode(h0[i]), hw_decode(h1[i])))
    bad_total += int(bad.sum())
print("total anomalous waves: %d" % bad_total)
