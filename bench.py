#!/usr/bin/env python3
"""bench.py — path-samples/s of the MI355X path-trace hot path (BASELINE.json metric).

A "step" = one pass of the hot path over one batch: ONE srt_render launch that traces
`spp` samples for every pixel of the frame (accumulator in registers, one framebuffer +
accumulator store per pixel).

N = 1 workload = BASELINE.json configs[1]: Scenes/Scene1.json, 1920x1080, 32 spp,
8 bounces, camera at origin, FOV 55, seed 0, inputs resident in HBM.
N > 1: the same frame, row-striped over the ranks in memory-row space with
32*N spp (per-GPU path-samples fixed -> weak scaling), joined by ONE gather to rank 0
over RCCL (software-raytracer_amd/stripes.py).  Launched by the driver as
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  Extra objects:
  roofline       HBM view the metric asks for: algorithmic bytes/launch ÷ kernel time vs 8 TB/s
  roofline_valu  the bound that actually limits this kernel (fp32 VALU, no FMA credit)
  cpu_baseline   the oracle (CPU port of the reference loop) on this box's host cores, on a
                 bounded sample of the same workload; also used to assert parity in-run.
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

WIDTH, HEIGHT, SPP, BOUNCES, SEED, FOV = 1920, 1080, 32, 8, 0, 55
SCENE = "Scene1"
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_LANEOPS = 256 * 128 * 2.4e9      # 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz, no FMA credit


def algorithmic_bytes(width, rows, n_objects, resume):
    """compulsory HBM bytes per launch = P*(4 B packed + 16 B accumulator write
    [+16 B read when resuming]) + scene image."""
    px = width * rows
    return px * (4 + 16 + (16 if resume else 0)) + n_objects * 64


def algorithmic_laneops_per_sample(rbar, n_sph, n_box):
    """F = R*(24*N_sph + 35*N_box) + 60*R + 30 fp32 lane-ops per path-sample."""
    return rbar * (24 * n_sph + 35 * n_box) + 60 * rbar + 30


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default=SCENE)
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--spp", type=int, default=SPP, help="samples per pixel per GPU-share (x N for N GPUs)")
    ap.add_argument("--bounces", type=int, default=BOUNCES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=16, help="spp of the bounded CPU sample (16 spp at 1080p = about 20 s of CPU work)")
    ap.add_argument("--balance", default="cost", choices=["cost", "equal"], help="row-stripe split for N>1")
    ap.add_argument("--gather", default="p2p", choices=["p2p", "padded"],
                    help="how cost-balanced (unequal) bands are joined: one grouped isend/irecv in place, or one padded dist.gather")
    ap.add_argument("--mesh", type=int, default=0, metavar="N",
                    help="EXTENSION workload (BASELINE configs[3]): replace Scene1's big ball by an N x N lat-long "
                         "tessellation (224 -> 99,904 triangles); implies --no-cpu-baseline")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

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

    W, H = args.width, args.height
    spp = args.spp * world
    scene_file = os.path.join(ROOT, "software-raytracer_amd", "scenes", args.scene + ".json")
    if args.mesh:
        import tempfile

        sj = json.load(open(scene_file))
        sj["SceneObjects"][64]["Renderer"] = {"Type": "Mesh", "Primitive": "UVSphere", "Radius": 1.0, "Stacks": args.mesh, "Slices": args.mesh}
        tmp = tempfile.NamedTemporaryFile("w", suffix=".json", delete=False)
        json.dump(sj, tmp)
        tmp.close()
        scene_file = tmp.name
        args.no_cpu_baseline = True
    scene = srt.host.Scene(scene_file)
    if scene.error:
        sys.exit("scene: " + scene.error)
    objs, n_obj = scene.objects_copy()
    meshes, n_mesh = scene.meshes()
    n_tri = sum(int(meshes[i].triangle_count) for i in range(n_mesh))
    n_sph = sum(1 for i in range(n_obj) if objs[i].type == srt.capi.OBJ_SPHERE)
    n_box = sum(1 for i in range(n_obj) if objs[i].type == srt.capi.OBJ_BOX)

    pt = srt.PathTracer(W, H, device=local_rank)
    pt.set_meshes(meshes, n_mesh)
    pt.set_scene(objs, n_obj)
    pt.set_camera(srt.default_camera(FOV))
    # render straight into a torch tensor so the gather needs no staging copy
    frame = torch.zeros((H, W), dtype=torch.int32, device=dev)
    pt.bind_output(d_framebuffer=frame.data_ptr())
    # kernel + gather on ONE non-default stream: ordered without host syncs (RCCL work is
    # enqueued behind the current stream's work)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    pt.set_stream(stream.cuda_stream)

    # ---- row stripes -----------------------------------------------------------------
    if world > 1 and args.balance == "cost":
        # per-row cost probe: rays per memory row from a 1-spp pass (deterministic, same on all ranks)
        row_cost = []
        probe = srt.PathTracer(W, H, device=local_rank)
        probe.set_meshes(meshes, n_mesh)
        probe.set_scene(objs, n_obj)
        probe.set_camera(srt.default_camera(FOV))
        band = 8
        for rb in range(0, H, band):
            re = min(rb + band, H)
            probe.render(spp=1, bounces=args.bounces, seed=SEED, rows=(rb, re), count_rays=True)
            # cost model: secondary rays dominate; +0.05 per pixel of fixed work
            st = probe.stats()
            c = (st.rays - W * (re - rb)) + 0.05 * W * (re - rb)
            row_cost += [c / (re - rb)] * (re - rb)
        probe.close()
        bands = stripes.partition_rows(H, world, row_cost, align=8)
        # refine with MEASURED kernel times (untimed set-up, like building an acceleration structure):
        # each band's row costs are rescaled so that the band's total matches its measured time, then
        # the rows are re-partitioned.  A few rounds converge; every rank computes the same split.
        cal_spp = spp  # the real launch: the library picks tiling / sample chunking from rows and spp
        best_bands, best_max = bands, float("inf")
        for _ in range(8):
            a, b = bands[rank]
            pt.render(spp=cal_spp, bounces=args.bounces, seed=SEED, first_sample=1, reset=True, rows=(a, b))
            t_loc = torch.tensor([pt.stats().kernel_ms], dtype=torch.float64, device=cdev)
            t_all = [torch.zeros_like(t_loc) for _ in range(world)]
            dist.all_gather(t_all, t_loc)
            times = [float(t.item()) for t in t_all]
            if max(times) < best_max:  # keep the best MEASURED split (identical decision on every rank)
                best_bands, best_max = bands, max(times)
            if max(times) <= 1.02 * (sum(times) / world):
                break
            for k, (x, y) in enumerate(bands):
                tot = sum(row_cost[x:y]) or 1.0
                f = times[k] / tot
                for r in range(x, y):
                    row_cost[r] *= f
            bands = stripes.partition_rows(H, world, row_cost, align=1)
        bands = best_bands
    else:
        bands = stripes.partition_rows(H, world)
    rb, re = bands[rank]

    # one 8-row launch so that code-object loading is not billed to the first step when --warmup is 0
    # (initialisation like srt_create / srt_set_scene, not a step)
    pt.render(spp=1, bounces=1, seed=SEED, first_sample=1, reset=True, rows=(0, min(8, H)))
    pt.wait()

    host_frame = torch.zeros((H, W), dtype=torch.int32) if rehearsal else None

    def step(count_rays=False):
        pt.render(spp=spp, bounces=args.bounces, seed=SEED, first_sample=1, reset=True, rows=(rb, re), count_rays=count_rays)
        if rehearsal and world > 1:  # gloo cannot move device memory: stage through the host
            stream.synchronize()
            host_frame[rb:re].copy_(frame[rb:re])
            stripes.gather_bands(host_frame, bands, rank, world, dist, method=args.gather)
        else:
            stripes.gather_bands(frame, bands, rank, world, dist, method=args.gather)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    try:
        for _ in range(max(args.warmup, 1) if world > 1 else args.warmup):
            step()
        fence()
    except RuntimeError as e:  # a backend that rejects the grouped point-to-point form
        if world > 1 and args.gather == "p2p":
            args.gather = "padded"
            if rank == 0:
                print("bench.py: grouped isend/irecv failed (%s); using the padded gather" % str(e).splitlines()[0], file=sys.stderr)
            for _ in range(max(args.warmup, 1)):
                step()
            fence()
        else:
            raise
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # per-launch kernel time from HIP events on the launch stream (one extra, untimed, launch
    # per sample so the event pair is read without perturbing the timed region)
    rays = 0
    for _ in range(min(args.steps, 5)):
        pt.render(spp=spp, bounces=args.bounces, seed=SEED, first_sample=1, reset=True, rows=(rb, re), count_rays=True)
        st = pt.stats()
        kernel_ms.append(st.kernel_ms)
        rays = st.rays
    k_ms = sum(kernel_ms) / len(kernel_ms)
    local_samples = W * (re - rb) * spp
    rbar_local = rays / local_samples

    total_samples = W * H * spp
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
            chk = srt.PathTracer(W, H, device=local_rank)
            chk.set_meshes(meshes, n_mesh)
            chk.set_scene(objs, n_obj)
            chk.set_camera(srt.default_camera(FOV))
            chk.render(spp=spp, bounces=args.bounces, seed=SEED)
            stripe_parity = bool(np.array_equal(chk.framebuffer(), gathered))
            chk.close()

    out = None
    if rank == 0:
        rbar = sum(p[1] for p in per_rank) / sum(p[2] for p in per_rank)
        abytes = algorithmic_bytes(W, re - rb, n_obj, resume=False)
        achieved_gbs = abytes / (k_ms * 1e-3) / 1e9
        lane_ops = algorithmic_laneops_per_sample(rbar_local, n_sph, n_box) * local_samples
        achieved_valu = lane_ops / (k_ms * 1e-3)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and world == 1:
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == "%s %dx%d spp%d b%d" % (args.scene, W, H, spp, args.bounces):
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "path-samples/sec at 1920x1080x8-bounce; achieved HBM GB/s vs peak",
            "value": value,
            "unit": "path-samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (shipped %s.json scene, counter-based RNG seed 0)" % args.scene,
            "config": {
                "workload": "%s.json %dx%d, %d spp (%d per GPU-share), %d bounces, FOV %d, camera at origin" %
                            (args.scene, W, H, spp, args.spp, args.bounces, FOV),
                "objects": {"spheres": n_sph, "boxes": n_box, "mesh_triangles": n_tri},
                "partition": "single frame" if world == 1 else "row stripes in memory-row space, %s split, one RCCL gather (%s)" % (args.balance, args.gather),
                "bands": bands,
                "rays_per_sample": rbar,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel": "srt::pathtrace_kernel",
                "kernel_ms": k_ms,
                "algorithmic_bytes_per_launch": abytes,
                "note": "compulsory bytes only (20 B/pixel + scene); the kernel is VALU-bound, see roofline_valu",
            },
            "roofline_valu": {
                "bound": "valu_fp32_no_fma",
                "achieved": achieved_valu / 1e12,
                "peak": VALU_PEAK_LANEOPS / 1e12,
                "unit": "T lane-op/s",
                "frac": achieved_valu / VALU_PEAK_LANEOPS,
                "algorithmic_laneops_per_sample": algorithmic_laneops_per_sample(rbar_local, n_sph, n_box),
                "rays_per_sample": rbar_local,
            },
            "per_rank": [{"kernel_ms": p[0], "rays_per_sample": p[1] / p[2], "rows": list(bands[i])} for i, p in enumerate(per_rank)],
        }
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
        cspp = args.cpu_spp
        oarr = C.cast(objs, C.POINTER(O.Object))
        t1 = time.perf_counter()
        ofb, oacc, orays = O.render(oarr, n_obj, O.default_environment(), O.default_camera(FOV), W, H, spp=cspp,
                                    bounces=args.bounces, seed=SEED, pow_mode=O.POW_SHARED, threads=cores)
        cpu_dt = time.perf_counter() - t1
        # same sample on the GPU: frame hash must match before any speed is reported
        pt.bind_output()  # own buffers
        pt.render(spp=cspp, bounces=args.bounces, seed=SEED, count_rays=True)
        gfb, gacc = pt.framebuffer(), pt.accumulator()
        parity = bool(np.array_equal(gfb, ofb) and np.array_equal(gacc.view(np.uint32), oacc.view(np.uint32)))
        # reference-faithful split (16 column stripes, Raytracer.cpp:330-342), 1 spp
        t2 = time.perf_counter()
        O.render(oarr, n_obj, O.default_environment(), O.default_camera(FOV), W, H, spp=1, bounces=args.bounces,
                 seed=SEED, pow_mode=O.POW_LIBM, threads=16, split=O.SPLIT_REF_COLS)
        ref_dt = time.perf_counter() - t2
        cpu_model = ""
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        out["cpu_baseline"] = {
            "value": W * H * cspp / cpu_dt,
            "unit": "path-samples/s",
            "cores": cores,
            "kind": "port",
            "sample": "same scene/camera/seed at %dx%d, %d spp, %d bounces, rows interleaved over %d threads (%.2f s)" %
                      (W, H, cspp, args.bounces, cores, cpu_dt),
            "cpu_model": cpu_model,
            "reference_split_16_column_stripes": {"value": W * H / ref_dt, "threads": 16, "spp": 1, "seconds": ref_dt},
            "parity_frame_hash_gpu": O.frame_hash(gfb),
            "parity_frame_hash_cpu": O.frame_hash(ofb),
            "parity_bit_exact": parity,
        }
        if not parity:
            out["value"] = None
            out["error"] = "GPU frame differs from the CPU oracle on the baseline sample; speed not reported"

    if rank == 0:
        print(json.dumps(out))
    pt.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
