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
                 `achieved` = EXECUTED lane-operations per launch / kernel time, and the executed work is COUNTED BY
                 THE RUN ITSELF: one extra, untimed launch of the same shape with SRT_RENDER_COUNT_WORK returns what the
                 kernel's loops did (pool steps, sphere / bound / box / BVH-child / triangle tests, srt_get_work_counts);
                 bench.py prices every test with SURVEY §8d's constants (24 / 35 / 40 lane-ops per sphere / box /
                 triangle test) plus three stated figures (VALU_MODEL below) and divides by the time and
                 the peak.  Executed instructions cannot exceed the issue capacity, so `frac` <= 1.  `counted` holds
                 the raw counts, so the fraction can be recomputed from the line alone.
  algorithmic    (outside `roofline`) the brute-force-equivalent lane-ops of the §8d formula over the same time: what a
                 kernel without culling and without the once-per-pixel primary hit would have had to execute —
                 `speedup_vs_bruteforce` = that / executed.  Round 3 printed this as `roofline.frac` (1.31).
  roofline_hbm   the HBM view the metric's wording asks for (compulsory bytes / kernel time vs 8 TB/s).
  readback       the frame's way into host memory (where the reference's frame ends: renderSurface->pixels,
                 Raytracer.cpp:64,75): srt_read_framebuffer into pinned and pageable memory, `value_including_readback`,
                 and a double-buffered loop that copies frame k while frame k + 1 renders.  Never in `value`.
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
# What a counted unit of executed work costs in fp32 VALU lane-operations (= instructions x lanes; an FMA is ONE issue slot, like
# every instruction, against a peak without FMA credit).  sphere / box / triangle are SURVEY §8d's constants — and the kernel's own
# instruction counts for the candidate test of a sphere (24: srt_kernel.hip.h part1) and the slab test; the others are stated here
# and derived in DESIGN.md §4.7: a cluster-bound test is 14 instructions (three differences, three 3-term FMA chains, the inflated
# radius, compare, mask), a quantized BVH child box 21 (six byte conversions, six FMAs, max3 / min3, the culling compares);
# `step` is everything a pool step runs besides those tests — ray generation (four hash rounds, two normalisations), shading with
# the environment's powf, the ordered fold, the task hand-out, the compaction and merge around the exact rounds, the second halves
# of the sphere tests, hit point and normal — per lane and step; `bvh_round` / `mesh_phase` what a node round and a traversal
# phase run besides their child and triangle tests (pops, shuffles, the scan, pushes; the root tests and the winner's fetch).
# `step` is calibrated ONCE on config 2, `bvh_round` and `mesh_phase` on config 4: (SQ_INSTS_VALU - the priced tests) / the count,
# from profiles/r04/pmc_c2.json and pmc_c4.json; profiles/r04/valu_model.json checks the model against the instruction counter of
# every other workload.
VALU_MODEL = {"sphere": 24, "box": 35, "triangle": 40, "bound": 14, "bvh_child": 21, "step": 780, "bvh_round": 240, "mesh_phase": 300, "sample": 30}

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


def algorithmic_laneops_per_sample(rbar, n_sph, n_box, n_tri=0):
    """F = R*(24*N_sph + 35*N_box + 40*N_tri) + 60*R + 30 fp32 lane-ops per path-sample — the scan of EVERY primitive
    for every ray (for the 99,904-triangle scenes: what the BVH saves is in this figure too)."""
    return rbar * (24 * n_sph + 35 * n_box + 40 * n_tri) + 60 * rbar + 30


def executed_laneops(wc, samples, mesh):
    """Lane-operations the launch EXECUTED, from its own loop counts (srt_work_counts as a dict) priced with VALU_MODEL:
    every test count is lane-level and executed (a wave runs a loop trip for all 64 lanes), a pool step, a BVH node round and a
    traversal phase cost their fixed parts for 64 lanes, every path-sample its accumulate / tone-map share.  None when the launch
    could not count."""
    if not wc or not wc.get("valid"):
        return None
    m = VALU_MODEL
    return (m["sphere"] * (wc["uniform_sphere_tests"] + wc["cluster_sphere_tests"]) + m["box"] * wc["box_tests"] + m["bound"] * wc["cluster_bound_tests"] +
            m["bvh_child"] * wc["bvh_child_tests"] + m["triangle"] * wc["triangle_tests"] + m["step"] * 64 * wc["pool_steps"] +
            m["bvh_round"] * 64 * wc["bvh_node_rounds"] + m["mesh_phase"] * 64 * wc["mesh_phases"] + m["sample"] * samples)


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
        # (the probe IS a launch — a whole frame's pool over a quarter of the pixels for 32 samples, about 8 sample-frames of work on
        # every rank — and it is outside the timed region: probe_ms says what it cost, value_including_probe_once what the
        # run's rate is with it paid once, so the balanced and the equal split can be compared end to end)
        calibration = {"calibration_launches": 1, "probe_launches": 1, "probe_ms": (time.perf_counter() - t_cal) * 1e3,
                       "calibration": "srt_estimate_row_costs: one device-side probe launch per rank — the path pool over a quarter of the pixels for the frame's first 32 samples, counting loop trips, no timing; the same numbers on every rank, no collective"}
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
    # the padded gather's buffers, made once (stripes.GatherBuffers: the sending side is a view of the frame)
    gather_frame = host_frame if rehearsal else frame
    gather_buffers = stripes.GatherBuffers(gather_frame, bands, rank, world) if world > 1 and gather_method == "padded" else None

    in_timed_region = [False]
    kernel_events = []  # (begin, end, gather end) HIP events on the launch stream around every TIMED step's srt_render and its gather

    def step(count_rays=False, timed=False, count_work=False):
        # (N = 1: no per-step events — the one pair around the whole timed region gives the time per step, and every event record is
        # a marker on the stream the next launch queues behind: three of them per step cost 1.5 % of config 2's step.  N > 1 needs the
        # split between kernel and gather.)
        timed = timed and world > 1
        if timed:
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(stream)
        # (the steps of the timed region carry no timing events of the library's either, SRT_RENDER_NO_TIMING: the region is timed from
        # outside; the launch after it is bracketed by the library's own pair: kernel_ms_last_timed_launch)
        pt.render(spp=spp, bounces=bounces, seed=SEED, first_sample=1, reset=True, rows=(rb, re), count_rays=count_rays, count_work=count_work,
                  timing=not in_timed_region[0])
        if timed:
            e1.record(stream)
        if world > 1:
            if rehearsal:  # gloo cannot move device memory: stage through the host
                stream.synchronize()
                host_frame[rb:re].copy_(frame[rb:re])
            stripes.gather_bands(gather_frame, bands, rank, world, dist, method=gather_method, buffers=gather_buffers)
        if timed:
            e2.record(stream)
            kernel_events.append((e0, e1, e2))

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
    in_timed_region[0] = True
    for _ in range(args.steps):
        step(timed=True)
    in_timed_region[0] = False
    ev1.record(stream)
    fence()
    dt = time.perf_counter() - t0
    stream_ms = ev0.elapsed_time(ev1) / args.steps  # per step on the launch stream (kernels + gather enqueue)
    step()                                           # one more identical step, outside the timed region, with the library's own event pair:
    last_launch_ms = float(pt.stats().kernel_ms)     # the kernel time of a steady-state launch as srt_get_stats reports it
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    launch_ms_mean = sum(a.elapsed_time(b) for a, b, _ in kernel_events) / len(kernel_events) if kernel_events else stream_ms  # this rank's srt_render launches only
    gather_ms_mean = sum(b.elapsed_time(c) for _, b, c in kernel_events) / len(kernel_events) if kernel_events else 0.0  # ... and its gathers (stream time behind the kernel)
    k_ms = stream_ms if world == 1 else launch_ms_mean
    st_timed = pt.stats()  # the shape of the last timed launch

    # rays per sample and the executed work (both deterministic): one extra, untimed launch of the same shape with the counters on
    step(count_rays=True, count_work=True)
    st_last = pt.stats()
    rays = st_last.rays
    wc = pt.work_counts().as_dict()
    counting_launch_ms = float(st_last.kernel_ms)
    shape = {"tile_rows": int(st_last.tile_rows), "grid_layers": int(st_last.sample_chunks), "chunk_samples": int(st_last.chunk_samples),
             "shape_source": "work record of the band's first launch" if st_last.shape_source else "static rule (request and grid)",
             "same_as_timed_launches": (st_last.tile_rows, st_last.sample_chunks, st_last.chunk_samples) == (st_timed.tile_rows, st_timed.sample_chunks, st_timed.chunk_samples)}
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

    ex_local = executed_laneops(wc, local_samples, bool(n_tri))
    mine = [k_ms, float(rays), float(local_samples), float(ex_local if ex_local is not None else -1.0), gather_ms_mean,
            float(wc["pool_steps"]), float(st_last.sample_chunks), float(st_last.tile_rows)]
    if world > 1:
        info = torch.tensor(mine, dtype=torch.float64, device=cdev)
        allinfo = [torch.zeros_like(info) for _ in range(world)]
        dist.all_gather(allinfo, info)
        per_rank = [[float(x) for x in t.tolist()] for t in allinfo]
    else:
        per_rank = [mine]

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

    # ---- the frame's way into host memory (N = 1): where the reference's frame ends (renderSurface->pixels, Raytracer.cpp:64,75).
    # After the timed region, never in `value`: srt_read_framebuffer into pinned and into pageable memory, and a double-buffered
    # loop — frame k is copied on a second stream while frame k + 1 renders into the other framebuffer (srt_bind_output does
    # not wait) — the way a host that blits every frame would run it.
    readback = None
    if world == 1 and rank == 0:
        import numpy as np

        step_ms = dt / args.steps * 1e3
        pinned = torch.empty((H, W), dtype=torch.int32).pin_memory()
        pageable = np.empty((H, W), dtype=np.uint32)

        def read_ms(ptr, reps=5):
            best = None
            for _ in range(reps):
                t = time.perf_counter()
                pt._ck(pt.L.srt_read_framebuffer(pt._h, C.c_void_p(ptr), W * 4, rb, re))
                e = (time.perf_counter() - t) * 1e3
                best = e if best is None else min(best, e)
            return best

        pin_ms = read_ms(pinned.data_ptr() + rb * W * 4)
        page_ms = read_ms(pageable.ctypes.data + rb * W * 4)
        frames = [frame, torch.zeros_like(frame)]
        hosts = [pinned, torch.empty((H, W), dtype=torch.int32).pin_memory()]
        copy_stream = torch.cuda.Stream(dev)
        copied = [torch.cuda.Event() for _ in range(2)]
        n_ov = max(args.steps, 4)

        def overlapped(n):
            for k in range(n):
                i = k & 1
                if k >= 2:
                    stream.wait_event(copied[i])  # the copy of frame k - 2 has left this framebuffer
                pt.bind_output(d_framebuffer=frames[i].data_ptr())
                pt.render(spp=spp, bounces=bounces, seed=SEED, first_sample=1, reset=True, rows=(rb, re))
                # srt_read_framebuffer_async: the copy of THIS frame on the copy stream, behind the render just enqueued
                pt.read_framebuffer_async(hosts[i].data_ptr() + rb * W * 4, rows=(rb, re), copy_stream=copy_stream.cuda_stream)
                copied[i].record(copy_stream)

        overlapped(2)
        torch.cuda.synchronize(dev)
        t_ov = time.perf_counter()
        overlapped(n_ov)
        torch.cuda.synchronize(dev)
        ov_ms = (time.perf_counter() - t_ov) * 1e3 / n_ov
        same = bool(torch.equal(hosts[0][rb:re], hosts[1][rb:re]) and torch.equal(hosts[0][rb:re], frames[0][rb:re].cpu()))
        pt.bind_output(d_framebuffer=frame.data_ptr())
        readback = {
            "frame_to_host_ms": pin_ms, "frame_to_host_ms_pageable": page_ms, "bytes": W * (re - rb) * 4,
            "how": "srt_read_framebuffer (synchronous hipMemcpy2D) of this launch's rows into pinned / pageable host memory, best of 5, after the timed region",
            "fraction_of_a_step": pin_ms / step_ms,
            "value_including_readback": total_samples / ((step_ms + pin_ms) * 1e-3),
            "ms_per_step_with_overlapped_readback": ov_ms,
            "value_with_overlapped_readback": total_samples / (ov_ms * 1e-3),
            "overlapped_how": "%d steps, two framebuffers: frame k is copied device-to-pinned on a second stream (srt_read_framebuffer_async) while frame k + 1 renders (srt_bind_output does not wait); wall clock / steps" % n_ov,
            "overlapped_frames_identical": same,
        }

    # ---- the first launch of a frame with WARM clocks (N = 1): a fresh context — no recorded costs, the dispatch order estimated on
    # the device — renders the same frame right after the timed region; next to kernel_ms_cold_first_launch (the run's very first
    # step, which also pays the clock ramp of a GPU that was idle a moment ago) this separates what the missing record costs
    # from what the power management does.  Event pair on the stream around srt_render, order estimate included.
    warm_first_ms = None
    if world == 1 and rank == 0:
        trials = []
        for _ in range(2):  # (two fresh contexts, the smaller figure: a lone outlier of tens of milliseconds has been seen on this pool)
            fresh = new_tracer()
            fresh.bind_output(d_framebuffer=frames[1].data_ptr())
            fresh.set_stream(stream.cuda_stream)
            # (what a context allocates once, on its first launch of a grid — the cost-record buffers, 1 MB of pinned memory at 1080p,
            # tens of milliseconds of host time — is initialisation like srt_create: a first pass allocates, then srt_set_scene makes
            # the context forget everything it has learned about the frame, as after any scene change)
            fresh.render(spp=4, bounces=1, seed=SEED, first_sample=1, reset=True, rows=(rb, re))
            fresh.wait()
            fresh.set_meshes(meshes, n_mesh)
            fresh.set_scene(objs, n_obj)
            for _ in range(3):
                pt.render(spp=spp, bounces=bounces, seed=SEED, first_sample=1, reset=True, rows=(rb, re))
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            f0.record(stream)
            fresh.render(spp=spp, bounces=bounces, seed=SEED, first_sample=1, reset=True, rows=(rb, re))
            f1.record(stream)
            torch.cuda.synchronize(dev)
            trials.append(f0.elapsed_time(f1))
            fresh.close()
        warm_first_ms = min(trials)

    out = None
    if rank == 0:
        rbar = sum(p[1] for p in per_rank) / sum(p[2] for p in per_rank)
        key = workload_key(cfg["scene"], cfg["mesh"], W, H, (rb, re), spp, bounces)
        all_counters = load_json(os.path.join(ROOT, "profiles", "counters.json"))
        counters = all_counters.get(key, {}) if world == 1 else {}
        rank_rows = [tuple(bands[i]) if not share else (rb, re) for i in range(len(per_rank))]
        # EXECUTED lane-ops of every rank's launch, counted by that rank's own counting launch (executed_laneops); one rank -> its
        # launch; N ranks -> the sum over the SLOWEST rank's launch time against N GPUs' peak, every rank's own fraction in per_rank.
        # Next to it, outside the roofline: the brute-force-equivalent count of the SURVEY §8d formula with that rank's own rays per
        # sample (mesh scenes: the BVH work as counted) — what a kernel without culling would have had to execute.
        counted_ok = all(p[3] >= 0 for p in per_rank)
        rank_ex = [p[3] if p[3] >= 0 else 0.0 for p in per_rank]
        rank_bf = []
        for (k_i, rays_i, samples_i, *_rest) in per_rank:
            rank_bf.append(algorithmic_laneops_per_sample(rays_i / samples_i, n_sph, n_box, n_tri) * samples_i)
        slowest_ms = max(p[0] for p in per_rank)
        achieved_valu = sum(rank_ex) / (slowest_ms * 1e-3)
        bruteforce_valu = sum(rank_bf) / (slowest_ms * 1e-3)
        peak_valu = VALU_PEAK_LANEOPS * len(per_rank)
        abytes = algorithmic_bytes(W, re - rb, n_obj, resume=False)
        abytes += 32 * traced_samples  # sample-chunked launch: + 16 B written and 16 B read per traced sample (the fold's stream; an upper figure since the chained chunks)
        achieved_gbs = abytes / (k_ms * 1e-3) / 1e9
        kind = ("executed lane-ops COUNTED by this run (one untimed launch of the same shape with SRT_RENDER_COUNT_WORK; `counted`) priced with "
                "`valu_model`: tests x the SURVEY §8d constants + the stated per-bound / per-BVH-child / per-step figures (DESIGN.md §4.7)")
        if not counted_ok:
            kind = "UNCOUNTED: a launch whose scene image lives in HBM keeps no loop counts; `achieved` is 0"
        if world > 1:
            kind += "; N = %d: sum of all ranks' executed lane-ops / the slowest rank's launch time, peak = %d GPUs" % (world, world)
        # pipe utilisation as rocprofv3 saw it: computed INSIDE the counter pass (its own SQ_INSTS_VALU over its own launch time), copied from profiles
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
                # what srt_render chose for this rank's launch (srt_stats): a function of the request, the grid, the CU count and the
                # band's recorded loop counts — the same in every run
                "launch_shape": shape,
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
                "kernel_ms_last_timed_launch": last_launch_ms,  # (the launch right after the timed region: the timed ones carry no events)
                "kernel_ms_cold_first_launch": cold_ms,
                "kernel_ms_first_launch_warm_clocks": warm_first_ms,
                "kernel_ms_counting_launch": counting_launch_ms,  # the same launch through the instantiation that keeps the loop counts (untimed)
                "peak_note": "%sfp32 VALU without FMA credit: 256 CU x 4 SIMD x 32 lanes x 2.4 GHz (contraction is forbidden by the bit-exactness contract)" %
                             ("%d GPUs x " % world if world > 1 else ""),
                "achieved_kind": kind,
                "counted": wc if world == 1 else {"note": "rank 0's counts; every rank's executed lane-ops in per_rank", **wc},
                "valu_model": VALU_MODEL,
                "executed_laneops_per_sample": sum(rank_ex) / sum(p[2] for p in per_rank),
                # bounce rays (every ray but the samples' primary ones) per lane-step of the pool: how full its 64 lanes are
                "pool_lane_efficiency": (rays - local_samples) / (64.0 * wc["pool_steps"]) if wc.get("pool_steps") else None,
                "rays_per_sample": rbar,
                "measured_valu_issue_frac": measured_issue,
                "measured_valu_issue_frac_note": "from profiles/counters.json, NOT from this run: SQ_INSTS_VALU x 64 / the launch time of the rocprofv3 pass "
                                                 "that counted it / peak (valu_issue_pass_launch_ms); clock-independent twin: measured_valu_issue_frac_grbm" if measured_issue else None,
                "measured_valu_issue_frac_grbm": counters.get("valu_issue_frac_grbm"),
                "measured_pass_launch_ms": counters.get("valu_issue_pass_launch_ms"),
                "measured_source": counters.get("source"),
            },
            "algorithmic": {
                "bruteforce_equivalent_tflops": bruteforce_valu / 1e12,
                "speedup_vs_bruteforce": bruteforce_valu / achieved_valu if achieved_valu else None,
                "laneops_per_sample": sum(rank_bf) / sum(p[2] for p in per_rank),
                "note": "SURVEY §8d formula F = R(24 N_sph + 35 N_box + 40 N_tri) + 60 R + 30 with this run's rays per sample: what a scan of every primitive for every ray "
                        "would execute in the same time; NOT a utilisation (it passes the peak where the kernel culls, walks a BVH and shares the primary hit among a pixel's samples)",
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
                        (" + 32 B per traced sample of the chunked launch's sample buffer — what the launch moves if every chunk goes through the buffer; chunks that "
                         "find their tile's running mean at their own first sample fold their samples themselves (chained chunks, DESIGN.md §4.5), so `traffic` can come out "
                         "BELOW this figure" if chunks > 1 else ""),
            },
            "per_rank": [{"kernel_ms": p[0], "gather_ms": p[4] if world > 1 else None, "rays_per_sample": p[1] / p[2], "rows": list(rank_rows[i]),
                          "executed_laneops": rank_ex[i], "frac": rank_ex[i] / (p[0] * 1e-3) / VALU_PEAK_LANEOPS, "pool_steps": p[5],
                          "grid_layers": int(p[6]), "tile_rows": int(p[7])} for i, p in enumerate(per_rank)],
        }
        if readback:
            out["readback"] = readback
        if calibration:
            out["config"].update(calibration)
            out["config"]["value_including_probe_once"] = total_samples * args.steps / (dt + calibration["probe_ms"] * 1e-3)
        if world > 1:
            # the N > 1 step = this rank's launch + the one gather: both halves, as the stream saw them (event pairs)
            out["gather_ms"] = max(p[4] for p in per_rank)
            out["gather_ms_note"] = "stream time between the end of a rank's kernel and the end of its part of the gather (rank 0: + the unpack copies), mean over the timed steps, max over ranks; unverified on > 1 GPU until the driver's run"
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
