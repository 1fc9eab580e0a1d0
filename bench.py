#!/usr/bin/env python3
"""bench.py — path-samples/s of the MI355X path-trace hot path (BASELINE.json metric).

A "step" = one pass of the hot path over one batch: ONE srt_render call that traces `spp` samples
for every pixel of this rank's rows (accumulator in registers or, for sample-chunked launches, one
streaming fold; one framebuffer + accumulator store per pixel), plus — for N > 1 — the one gather.

Workloads (BASELINE.json `configs`, 1-based like SURVEY §8):
  N = 1 (default)      config 2: Scenes/Scene1.json, 1920x1080, 32 spp, 8 bounces, camera at origin, FOV 55.
  N > 1 (the driver)   config 3's frame and scene, row-striped in contiguous memory-row bands of equal ESTIMATED
                       cost (the library's device-side probe, srt_estimate_row_costs: deterministic, every rank
                       computes the same split, no collective, no calibration launch; --balance equal = bands of
                       equal height), 64*N spp (per-GPU path-samples fixed for N >= 2 -> "weak"; N = 8 is
                       exactly config 3: 512 spp), joined by ONE dist.gather to rank 0 over RCCL.
  --config C           any of configs 2..5 in its stated form (4 and 5 use the 99,904-triangle ball).
  --rank k/N           with --config: render rank k's share of an N-rank run (equal stripe, the
                       config's full spp) on ONE GPU — how configs 3 and 5 are exercised without an
                       8-GPU node.

Prints ONE JSON line on rank 0 and exits non-zero if an in-run parity check failed.  Extra objects:
  roofline       the bound that limits the kernel: fp32 VALU issue without FMA credit.
                 `achieved` = algorithmic lane-ops per launch / kernel time.  For analytic scenes the
                 lane-ops are the brute-force-equivalent count of the §8d formula (the kernel culls, so
                 this is an algorithmic-throughput fraction); for mesh scenes they are COUNTED (BVH node
                 and triangle tests per ray, profiles/mesh_counts.json).  `measured_valu_issue_frac` is
                 the pipe utilisation from rocprofv3 counters (profiles/counters.json), `traffic` the
                 HBM bytes per launch from the same file.
  roofline_hbm   the HBM view the metric's wording asks for (compulsory bytes / kernel time vs 8 TB/s).
  cpu_baseline   the oracle (CPU port of the reference loop) on this box's host cores, on a bounded
                 sample of the same workload; also used to assert parity in-run.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED, FOV = 0, 55
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_LANEOPS = 256 * 128 * 2.4e9      # 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz, no FMA credit
BAND_ALIGN = 2                             # rows: boundaries of the cost-balanced split (8 cost 2-3 points of balance at N = 8)
NODE_OPS, TRI_OPS = 240, 40                # lane-ops per BVH node item (8 child boxes) / triangle test

# BASELINE.json configs, 1-based.  mesh = tessellation of Scene1's big ball (224 -> 99,904 triangles)
CONFIGS = {
    1: dict(scene="Scene1", mesh=0, width=256, height=256, spp=1, bounces=4, ranks=1),
    2: dict(scene="Scene1", mesh=0, width=1920, height=1080, spp=32, bounces=8, ranks=1),
    3: dict(scene="Scene1", mesh=0, width=1920, height=1080, spp=512, bounces=8, ranks=8),
    4: dict(scene="Scene1", mesh=224, width=1920, height=1080, spp=64, bounces=8, ranks=1),
    5: dict(scene="Scene1", mesh=224, width=3840, height=2160, spp=1024, bounces=16, ranks=8),
}


def algorithmic_bytes(width, rows, n_objects, resume):
    """compulsory HBM bytes per launch = P*(4 B packed + 16 B accumulator write
    [+16 B read when resuming]) + scene image."""
    px = width * rows
    return px * (4 + 16 + (16 if resume else 0)) + n_objects * 64


def algorithmic_laneops_per_sample(rbar, n_sph, n_box, node_items=0.0, tri_tests=0.0):
    """F = R*(24*N_sph + 35*N_box) + 60*R + 30 fp32 lane-ops per path-sample, plus — for
    mesh scenes — the COUNTED BVH work per sample (node items x 240 + triangle tests x 40)."""
    return rbar * (24 * n_sph + 35 * n_box) + 60 * rbar + 30 + node_items * NODE_OPS + tri_tests * TRI_OPS


def workload_key(scene, mesh, width, height, rows, spp, bounces):
    """Key under which profiles/counters.json and profiles/mesh_counts.json file a workload."""
    return "%s%s %dx%d rows%d-%d spp%d b%d" % (scene, "+mesh%d" % mesh if mesh else "", width, height, rows[0], rows[1], spp, bounces)


def host_cpu_share():
    """CPUs this process may actually use: min(affinity, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def scene_file_for(scene, mesh):
    """Path of the scene JSON; mesh > 0 writes a temp copy with Scene1's big ball (object 64) replaced
    by an N x N lat-long tessellation (the EXTENSION workload of configs 4-5)."""
    path = os.path.join(ROOT, "software-raytracer_amd", "scenes", scene + ".json")
    if not mesh:
        return path
    import tempfile

    sj = json.load(open(path))
    sj["SceneObjects"][64]["Renderer"] = {"Type": "Mesh", "Primitive": "UVSphere", "Radius": 1.0, "Stacks": mesh, "Slices": mesh}
    tmp = tempfile.NamedTemporaryFile("w", suffix=".json", delete=False)
    json.dump(sj, tmp)
    tmp.close()
    TEMP_FILES.append(tmp.name)
    return tmp.name


TEMP_FILES = []


def load_json(path):
    try:
        return json.load(open(path))
    except (OSError, ValueError):
        return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 4, 5], help="BASELINE config in its stated form (default: 2 at N=1)")
    ap.add_argument("--rank", default="", metavar="k/N", help="with --config on ONE GPU: render rank k's equal stripe of an N-rank run")
    ap.add_argument("--scene", default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None, help="samples per pixel of the launch (N > 1 without --config: per GPU-share, x N)")
    ap.add_argument("--bounces", type=int, default=None)
    ap.add_argument("--mesh", type=int, default=None, metavar="N", help="EXTENSION: replace Scene1's big ball by an N x N tessellation")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=16, help="spp of the bounded CPU sample (16 spp at 1080p = about 20 s of CPU work)")
    ap.add_argument("--balance", default="probe", choices=["equal", "probe"],
                    help="row-stripe split for N > 1 (and for --rank): 'probe' (default): contiguous bands of equal estimated cost from the "
                         "library's device-side probe (srt_estimate_row_costs: the path pool over a quarter of the pixels for 32 samples, loop "
                         "trips counted, not timed — deterministic: no collective, no launch of the workload) + one dist.gather of bands padded "
                         "to the tallest; 'equal': bands of equal height + one in-place dist.gather (north_star's literal form; on "
                         "sky-over-floor scenes the slowest of 8 takes 2.2x the mean)")
    ap.add_argument("--gather", default="padded", choices=["padded", "p2p"],
                    help="cost-balanced (unequal) bands only: one dist.gather of bands padded to the tallest, or one grouped isend/irecv")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    # ---- workload ----------------------------------------------------------------------
    cfg_id = args.config or (2 if world == 1 else 3)
    cfg = dict(CONFIGS[cfg_id])
    if not args.config and world > 1:
        cfg["spp"] = 64 * world  # per-GPU path-samples fixed; 512 spp = config 3 at N = 8
    explicit_spp = args.spp is not None
    for key in ("scene", "width", "height", "spp", "bounces", "mesh"):
        if getattr(args, key) is not None:
            cfg[key] = getattr(args, key)
    if explicit_spp and world > 1 and not args.config:
        cfg["spp"] = args.spp * world
    W, H, spp, bounces = cfg["width"], cfg["height"], cfg["spp"], cfg["bounces"]
    share = None
    if args.rank:
        k, n = (int(v) for v in args.rank.split("/"))
        if world != 1 or not (0 <= k < n):
            sys.exit("--rank k/N emulates one rank of N on ONE GPU (0 <= k < N)")
        share = (k, n)

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the product path has no CPU fallback")
    # SRT_BENCH_REHEARSAL=1: every rank shares cuda:0 and the collectives run over gloo with
    # host staging — only to exercise the multi-rank control flow on a one-GPU box.
    rehearsal = os.environ.get("SRT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    cdev = torch.device("cpu") if rehearsal else dev  # where small collective payloads live

    srt = importlib.import_module("software-raytracer_amd")
    stripes = importlib.import_module("software-raytracer_amd.stripes")

    scene = srt.host.Scene(scene_file_for(cfg["scene"], cfg["mesh"]))
    for f in TEMP_FILES:  # (the mesh workloads' generated scene file: parsed, no longer needed)
        try:
            os.unlink(f)
        except OSError:
            pass
    if scene.error:
        sys.exit("scene: " + scene.error)
    objs, n_obj = scene.objects_copy()
    meshes, n_mesh = scene.meshes()
    n_tri = sum(int(meshes[i].triangle_count) for i in range(n_mesh))
    n_sph = sum(1 for i in range(n_obj) if objs[i].type == srt.capi.OBJ_SPHERE)
    n_box = sum(1 for i in range(n_obj) if objs[i].type == srt.capi.OBJ_BOX)

    def new_tracer():
        t = srt.PathTracer(W, H, device=local_rank)
        t.set_meshes(meshes, n_mesh)
        t.set_scene(objs, n_obj)
        t.set_camera(srt.default_camera(FOV))
        return t

    pt = new_tracer()
    # render straight into a torch tensor so the gather needs no staging copy
    frame = torch.zeros((H, W), dtype=torch.int32, device=dev)
    pt.bind_output(d_framebuffer=frame.data_ptr())
    # kernel + gather on ONE non-default stream: ordered without host syncs (RCCL work is
    # enqueued behind the current stream's work)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    pt.set_stream(stream.cuda_stream)

    # ---- row stripes -----------------------------------------------------------------
    calibration = None
    n_parts = share[1] if share else world
    if n_parts > 1 and args.balance == "probe":
        t_cal = time.perf_counter()
        row_cost = pt.estimate_row_costs(bounces, SEED)
        bands = stripes.partition_rows(H, n_parts, row_cost, align=BAND_ALIGN)
        calibration = {"calibration_launches": 0, "probe_launches": 1, "calibration_ms": (time.perf_counter() - t_cal) * 1e3,
                       "calibration": "none: srt_estimate_row_costs is one device-side probe — the path pool over a quarter of the pixels for the first 32 samples, counting loop trips, no timing; the same numbers on every rank"}
        rb, re = bands[share[0] if share else rank]
    elif share:
        bands = stripes.partition_rows(H, share[1])
        rb, re = bands[share[0]]
    else:
        bands = stripes.partition_rows(H, world)
        rb, re = bands[rank]
    equal_bands = len({b - a for a, b in bands}) == 1
    gather_method = "gather" if equal_bands else args.gather

    # one 8-row launch so that code-object loading is not billed to the first step when --warmup is 0
    # (initialisation like srt_create / srt_set_scene, not a step)
    pt.render(spp=1, bounces=1, seed=SEED, first_sample=1, reset=True, rows=(0, min(8, H)))
    pt.wait()

    host_frame = torch.zeros((H, W), dtype=torch.int32) if rehearsal else None

    kernel_events = []  # (begin, end) HIP events on the launch stream around every TIMED step's srt_render, the gather outside

    def step(count_rays=False, timed=False):
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
        pt.render(spp=spp, bounces=bounces, seed=SEED, first_sample=1, reset=True, rows=(rb, re), count_rays=count_rays)
        if timed:
            e1.record(stream)
            kernel_events.append((e0, e1))
        if world == 1:
            return
        if rehearsal:  # gloo cannot move device memory: stage through the host
            stream.synchronize()
            host_frame[rb:re].copy_(frame[rb:re])
            stripes.gather_bands(host_frame, bands, rank, world, dist, method=gather_method)
        else:
            stripes.gather_bands(frame, bands, rank, world, dist, method=gather_method)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- warm-up (W untimed steps).  The first one is the COLD launch of this frame: no block costs have
    # been recorded yet, the dispatch order is the host-derived one.  Its kernel time is reported next to
    # the steady-state time (a moving camera sees the cold number every frame).
    cold_ms = None
    n_warm = max(args.warmup, 1) if world > 1 else args.warmup
    for i in range(n_warm):
        step()
        if i == 0:
            cold_ms = float(pt.stats().kernel_ms)  # synchronises; warm-up only
    fence()

    # ---- timed region: exactly K steps, barrier + synchronize on both sides; HIP events on the launch
    # stream bracket the same region (kernel time per launch = event time / K at N = 1)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step(timed=True)
    ev1.record(stream)
    fence()
    dt = time.perf_counter() - t0
    stream_ms = ev0.elapsed_time(ev1) / args.steps  # per step on the launch stream (kernels + gather enqueue)
    last_launch_ms = float(pt.stats().kernel_ms)     # the library's own event pair around the LAST timed launch
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    launch_ms_mean = sum(a.elapsed_time(b) for a, b in kernel_events) / len(kernel_events)  # this rank's srt_render launches only
    k_ms = stream_ms if world == 1 else launch_ms_mean

    # rays per sample (deterministic): one extra, untimed launch with the counter on
    step(count_rays=True)
    st_last = pt.stats()
    rays = st_last.rays
    # sample-chunked launch (srt_stats.sample_chunks > 1): the colours of every TRACED sample go through HBM once (16 B
    # written by pathtrace_kernel, 16 B read by fold_kernel).  Traced pixels = pixels whose primary ray hits something:
    # a 1-spp / 1-bounce launch casts exactly one extra ray for each of them.
    chunks = int(st_last.sample_chunks)
    traced_samples = 0
    if chunks > 1:
        pt.render(spp=1, bounces=1, seed=SEED, first_sample=1, reset=True, rows=(rb, re), count_rays=True)
        traced_samples = (pt.stats().rays - W * (re - rb)) * spp
    local_samples = W * (re - rb) * spp
    rbar_local = rays / local_samples

    total_samples = (W * (re - rb) if share else W * H) * spp
    value = total_samples * args.steps / dt

    if world > 1:
        info = torch.tensor([k_ms, float(rays), float(local_samples)], dtype=torch.float64, device=cdev)
        allinfo = [torch.zeros_like(info) for _ in range(world)]
        dist.all_gather(allinfo, info)
        per_rank = [[float(x) for x in t.tolist()] for t in allinfo]
    else:
        per_rank = [[k_ms, float(rays), float(local_samples)]]

    # the gathered multi-rank frame must equal a single-device render of the whole frame
    stripe_parity = None
    if world > 1:
        step()
        fence()
        if rank == 0:
            import numpy as np

            gathered = (host_frame if rehearsal else frame.cpu()).numpy().view(np.uint32)
            chk = new_tracer()
            chk.render(spp=spp, bounces=bounces, seed=SEED)
            stripe_parity = bool(np.array_equal(chk.framebuffer(), gathered))
            chk.close()

    out = None
    if rank == 0:
        rbar = sum(p[1] for p in per_rank) / sum(p[2] for p in per_rank)
        key = workload_key(cfg["scene"], cfg["mesh"], W, H, (rb, re), spp, bounces)
        all_counters = load_json(os.path.join(ROOT, "profiles", "counters.json"))
        all_mesh_counts = load_json(os.path.join(ROOT, "profiles", "mesh_counts.json")) if n_tri else {}
        counters = all_counters.get(key, {}) if world == 1 else {}
        rank_rows = [tuple(bands[i]) if not share else (rb, re) for i in range(len(per_rank))]
        # lane-ops of every rank's launch (SURVEY §8d formula with that rank's own rays per sample; mesh scenes: + the counted
        # BVH work of that rank's band from profiles/mesh_counts.json), then: one rank -> its launch; N ranks -> the sum of all
        # ranks' lane-ops over the SLOWEST rank's launch time against N GPUs' peak, with every rank's own fraction in per_rank
        uncounted = False
        rank_ops, rank_f = [], []
        for (k_i, rays_i, samples_i), rows_i in zip(per_rank, rank_rows):
            mc = all_mesh_counts.get(workload_key(cfg["scene"], cfg["mesh"], W, H, rows_i, spp, bounces), {}) if n_tri else {}
            uncounted = uncounted or (bool(n_tri) and not mc)
            f_i = algorithmic_laneops_per_sample(rays_i / samples_i, n_sph, n_box, mc.get("node_items_per_sample", 0.0), mc.get("triangle_tests_per_sample", 0.0))
            rank_f.append(f_i)
            rank_ops.append(f_i * samples_i)
        slowest_ms = max(p[0] for p in per_rank)
        achieved_valu = sum(rank_ops) / (slowest_ms * 1e-3)
        peak_valu = VALU_PEAK_LANEOPS * len(per_rank)
        abytes = algorithmic_bytes(W, re - rb, n_obj, resume=False)
        abytes += 32 * traced_samples  # sample-chunked launch: + 16 B written and 16 B read per traced sample (the fold's stream)
        achieved_gbs = abytes / (k_ms * 1e-3) / 1e9
        if uncounted:
            kind = "UNCOUNTED: no profiles/mesh_counts.json entry for this workload; the BVH work is missing from `achieved`"
        elif n_tri:
            kind = "counted: analytic part by the SURVEY §8d formula + BVH node items x %d + triangle tests x %d (profiles/mesh_counts.json)" % (NODE_OPS, TRI_OPS)
        else:
            kind = "brute-force-equivalent lane-ops of the SURVEY §8d formula (the kernel culls: algorithmic throughput, not pipe utilisation)"
        if world > 1:
            kind += "; N = %d: sum of all ranks' lane-ops / the slowest rank's launch time, peak = %d GPUs" % (world, world)
        # pipe utilisation: computed INSIDE the counter pass (its own SQ_INSTS_VALU over its own launch time), copied from profiles
        measured_issue = counters.get("valu_issue_frac")
        custom = any(getattr(args, k) is not None for k in ("scene", "width", "height", "spp", "bounces", "mesh"))
        what = "custom workload (config %d with overrides)" % cfg_id if custom else "config %d" % cfg_id
        if world > 1 and not args.config and not custom:
            what = "config 3's frame at 64 spp per GPU-share" + (" = config 3" if world == 8 else "")
        if share:
            what += ", rank %d of %d's share on one GPU (memory rows %d-%d)" % (share[0], share[1], rb, re)
        out = {
            "metric": "path-samples/sec at 1920x1080x8-bounce; achieved HBM GB/s vs peak",
            "value": value,
            "unit": "path-samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            # per-GPU work fixed as N grows (the default N > 1 sequence: 64 spp per GPU-share) -> weak; a BASELINE config in
            # its stated form holds the frame AND the samples fixed -> strong
            "scaling": "strong" if (args.config and world > 1) else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (shipped %s.json scene%s, counter-based RNG seed 0)" %
                    (cfg["scene"], ", big ball as a %d-triangle mesh" % n_tri if n_tri else ""),
            "config": {
                "workload": "BASELINE %s: %s.json %dx%d, %d spp, %d bounces, FOV %d, camera at origin" %
                            (what, cfg["scene"], W, H, spp, bounces, FOV),
                "workload_key": key,
                "objects": {"spheres": n_sph, "boxes": n_box, "mesh_triangles": n_tri},
                "partition": "single frame" if world == 1 and not share else
                             "row stripes in memory-row space, %s bands, one %s" % (
                                 args.balance,
                                 "dist.gather over RCCL" if gather_method == "gather" else "RCCL gather (%s)" % gather_method),
                "bands": bands if not share else [[rb, re]],
                "rays_per_sample": rbar,
            },
            "roofline": {
                "bound": "valu",
                "achieved": achieved_valu / 1e12,
                "peak": peak_valu / 1e12,
                "unit": "TFLOP/s",
                "frac": achieved_valu / peak_valu,
                "traffic": counters.get("hbm_bytes_per_launch"),
                "kernel": "srt::pathtrace_kernel" + (" + srt::fold_kernel (%d sample chunks)" % chunks if chunks > 1 else ""),
                "kernel_ms": slowest_ms,
                "kernel_ms_source": "HIP events on the launch stream around the %d timed steps / %d" % (args.steps, args.steps)
                                    if world == 1 else "the SLOWEST rank's mean over the HIP event pairs around each timed step's srt_render (the gather outside); all ranks in per_rank",
                "kernel_ms_last_timed_launch": last_launch_ms,
                "kernel_ms_cold_first_launch": cold_ms,
                "peak_note": "%sfp32 VALU without FMA credit: 256 CU x 4 SIMD x 32 lanes x 2.4 GHz (contraction is forbidden by the bit-exactness contract)" %
                             ("%d GPUs x " % world if world > 1 else ""),
                "achieved_kind": kind,
                "algorithmic_laneops_per_sample": sum(rank_ops) / sum(p[2] for p in per_rank),
                "rays_per_sample": rbar,
                "measured_valu_issue_frac": measured_issue,
                "measured_valu_issue_frac_note": "from profiles/counters.json, NOT from this run: SQ_INSTS_VALU x 64 / the launch time of the rocprofv3 pass "
                                                 "that counted it / peak (valu_issue_pass_launch_ms); clock-independent twin: measured_valu_issue_frac_grbm" if measured_issue else None,
                "measured_valu_issue_frac_grbm": counters.get("valu_issue_frac_grbm"),
                "measured_pass_launch_ms": counters.get("valu_issue_pass_launch_ms"),
                "measured_source": counters.get("source"),
            },
            "roofline_hbm": {
                "bound": "hbm",
                "achieved": achieved_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS,
                "traffic": counters.get("hbm_bytes_per_launch"),
                "algorithmic_bytes_per_launch": abytes,
                "note": "compulsory bytes only (20 B/pixel + scene%s); not the limiting resource" %
                        (" + 32 B per traced sample of the chunked launch's sample buffer" if chunks > 1 else ""),
            },
            "per_rank": [{"kernel_ms": p[0], "rays_per_sample": p[1] / p[2], "rows": list(rank_rows[i]), "algorithmic_laneops": rank_ops[i],
                          "frac": rank_ops[i] / (p[0] * 1e-3) / VALU_PEAK_LANEOPS} for i, p in enumerate(per_rank)],
        }
        if calibration:
            out["config"].update(calibration)
        if world > 1 or share:
            # what every rank's launch of this split took when the ranks were run one after the other on ONE GPU (kernel times
            # only, no gather; tools/emulate_ranks.py, committed as profiles/emulated_ranks.json) — both splits, so that the
            # line of an N-GPU run can be read against it.  No scaling curve has been measured on more than one GPU.
            emu = load_json(os.path.join(ROOT, "profiles", "emulated_ranks.json")).get("config %d" % cfg_id, {})
            n_key = str(share[1] if share else world)
            if emu.get(n_key):
                out["config"]["emulated_on_one_gpu"] = dict(emu[n_key], note=emu.get("note"))
        if world > 1:
            out["stripe_parity_vs_single_device"] = stripe_parity
            if stripe_parity is False:
                out["value"] = None
                out["error"] = "gathered row stripes differ from the single-device frame; speed not reported"
        if rehearsal:
            out["rehearsal"] = "all ranks on cuda:0, gloo + host staging: control-flow test only, not a measurement"

    # ---- CPU baseline + in-run parity (rank 0, N = 1 only) -----------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import numpy as np
        import srt_oracle_py as O

        cores = host_cpu_share()
        oarr = C.cast(objs, C.POINTER(O.Object))
        omesh = (C.cast(meshes, C.POINTER(O.Mesh)), n_mesh) if n_mesh else None
        if n_tri:
            # the CPU port scans every triangle for every ray (~10^5 x the work of an analytic ray): the bounded
            # sample is a 32 x 16 pixel window over the mesh, inside this rank's rows, at the config's bounces
            cspp = min(args.cpu_spp, 4)
            cy = min(max(H // 2, (H - re) + 8), (H - rb) - 8)  # scene row nearest the image centre inside the band
            mr = H - 1 - cy
            rows_w, cols_w = (max(rb, mr - 8), min(re, mr + 8)), (W // 2 - 16, W // 2 + 16)
        else:
            cspp = args.cpu_spp
            rows_w, cols_w = (rb, re), None
        t1 = time.perf_counter()
        ofb, oacc, orays = O.render(oarr, n_obj, O.default_environment(), O.default_camera(FOV), W, H, spp=cspp,
                                    bounces=bounces, seed=SEED, pow_mode=O.POW_SHARED, threads=cores, rows=rows_w, cols=cols_w, meshes=omesh)
        cpu_dt = time.perf_counter() - t1
        cw = (cols_w[1] - cols_w[0]) if cols_w else W
        cpu_samples = cw * (rows_w[1] - rows_w[0]) * cspp
        # same sample on the GPU: the frames must match before any speed is reported
        pt.bind_output()  # own buffers
        pt.render(spp=cspp, bounces=bounces, seed=SEED, rows=(rb, re))
        gfb, gacc = pt.framebuffer(), pt.accumulator()
        cs = slice(*cols_w) if cols_w else slice(None)
        ys = slice(H - rows_w[1], H - rows_w[0])  # scene rows of the window (accumulator layout)
        g_win, o_win = gfb[rows_w[0]:rows_w[1], cs], ofb[rows_w[0]:rows_w[1], cs]
        parity = bool(np.array_equal(g_win, o_win) and np.array_equal(gacc[ys, cs].view(np.uint32), oacc[ys, cs].view(np.uint32)))
        ref_split = None
        if not n_tri and not share:
            # reference-faithful split (16 column stripes, Raytracer.cpp:330-342), 1 spp
            t2 = time.perf_counter()
            O.render(oarr, n_obj, O.default_environment(), O.default_camera(FOV), W, H, spp=1, bounces=bounces,
                     seed=SEED, pow_mode=O.POW_LIBM, threads=16, split=O.SPLIT_REF_COLS)
            ref_dt = time.perf_counter() - t2
            ref_split = {"value": W * H / ref_dt, "threads": 16, "spp": 1, "seconds": ref_dt}
        cpu_model = ""
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        out["cpu_baseline"] = {
            "value": cpu_samples / cpu_dt,
            "unit": "path-samples/s",
            "cores": cores,
            "kind": "port",
            "sample": "same scene/camera/seed, %s of the %dx%d frame, %d spp, %d bounces, over %d threads (%.2f s)" %
                      ("memory rows %d-%d x columns %d-%d" % (rows_w + cols_w) if cols_w else "memory rows %d-%d" % rows_w,
                       W, H, cspp, bounces, cores, cpu_dt),
            "cpu_model": cpu_model,
            "reference_split_16_column_stripes": ref_split,
            "parity_frame_hash_gpu": O.frame_hash(g_win),
            "parity_frame_hash_cpu": O.frame_hash(o_win),
            "parity_bit_exact": parity,
        }
        if not parity:
            out["value"] = None
            out["error"] = "GPU frame differs from the CPU oracle on the baseline sample; speed not reported"

    failed = False
    if rank == 0:
        print(json.dumps(out))
        failed = bool(out.get("error"))
    pt.close()
    if world > 1:
        flag = torch.tensor([1 if failed else 0], dtype=torch.int32, device=cdev)
        dist.broadcast(flag, src=0)
        failed = bool(flag.item())
        dist.destroy_process_group()
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
