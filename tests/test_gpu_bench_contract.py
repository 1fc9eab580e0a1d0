"""bench.py's output contract (one JSON line, the driver's keys, `roofline` and `cpu_baseline` objects),
on a small workload, and the multi-rank control flow rehearsed with two ranks sharing the GPU over gloo."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(stdout):
    lines = [l for l in stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--width", "480", "--height", "270",
                        "--spp", "4", "--cpu-spp", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "path-samples/s" and d["dtype"] == "f32" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 480 * 270 * 4 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma", "valu") and rf["unit"] in ("GB/s", "TFLOP/s") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and "traffic" in rf
    # round 4: the roofline is COUNTED by the run (executed tests priced with roofline.valu_model) and is a fraction <= 1 that can be
    # recomputed from the line alone; the brute-force-equivalent figure of round 3 lives outside it
    assert 0 < rf["frac"] <= 1.0 and rf["counted"]["valid"] == 1 and rf["counted"]["pool_steps"] > 0
    m, c = rf["valu_model"], rf["counted"]
    ops = (m["sphere"] * (c["uniform_sphere_tests"] + c["cluster_sphere_tests"]) + m["box"] * c["box_tests"] + m["bound"] * c["cluster_bound_tests"] +
           m["bvh_child"] * c["bvh_child_tests"] + m["triangle"] * c["triangle_tests"] + m["step"] * 64 * c["pool_steps"] +
           m["bvh_round"] * 64 * c["bvh_node_rounds"] + m["mesh_phase"] * 64 * c["mesh_phases"] + m["sample"] * 480 * 270 * 4)
    assert abs(ops / (rf["kernel_ms"] * 1e-3) / 1e12 - rf["achieved"]) / rf["achieved"] < 1e-9
    assert c["closest_hit_calls"] == c["pool_steps"] + c["waves"] and c["uniform_sphere_tests"] == c["closest_hit_calls"] * 3 * 64  # Scene1: 3 uniform spheres
    assert d["algorithmic"]["speedup_vs_bruteforce"] > 0 and "frac" not in d["algorithmic"]
    sh = d["config"]["launch_shape"]
    assert sh["tile_rows"] in (1, 2, 4, 8) and sh["grid_layers"] >= 1 and sh["same_as_timed_launches"] is True
    # the frame's way into host memory is reported, never part of `value`
    rb = d["readback"]
    assert rb["frame_to_host_ms"] > 0 and rb["frame_to_host_ms_pageable"] > 0 and rb["overlapped_frames_identical"] is True
    assert rb["value_including_readback"] < d["value"] and rb["ms_per_step_with_overlapped_readback"] > 0
    assert rf["kernel_ms_first_launch_warm_clocks"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == d["unit"] and cb["sample"]
    assert cb["parity_bit_exact"] is True  # the GPU frame and the oracle's frame of the same sample are identical


@pytest.mark.parametrize("balance,port", [("equal", 29571), ("probe", 29573)])
def test_two_rank_rehearsal(balance, port):
    """Two ranks sharing the GPU over gloo: equal bands + one dist.gather (the default), and bands of equal estimated
    cost from srt_estimate_row_costs (no collective, every rank computes the same split) + one padded dist.gather."""
    env = dict(os.environ, SRT_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--width", "640", "--height", "360", "--spp", "8", "--balance", balance], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["stripe_parity_vs_single_device"] is True
    bands = d["config"]["bands"]
    assert bands[0][0] == 0 and bands[-1][1] == 360 and bands[0][1] == bands[1][0]
    assert len(d["per_rank"]) == 2
    # the roofline of an N > 1 line is summed over the ranks: N GPUs' peak, all ranks' lane-ops over the slowest rank's launch time
    rf = d["roofline"]
    assert abs(rf["peak"] - 2 * 78.6432) < 1e-6 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert abs(rf["kernel_ms"] - max(p["kernel_ms"] for p in d["per_rank"])) < 1e-9
    ops = sum(p["executed_laneops"] for p in d["per_rank"])
    assert abs(rf["achieved"] * 1e12 - ops / (rf["kernel_ms"] * 1e-3)) / (rf["achieved"] * 1e12) < 1e-9
    assert all(0 < p["frac"] <= 1 for p in d["per_rank"]) and rf["frac"] <= 1
    # the step the driver times = kernel + the one gather: both halves are in the line (event pairs on the launch stream)
    assert d["gather_ms"] >= 0 and all(p["gather_ms"] is not None and p["gather_ms"] >= 0 for p in d["per_rank"])
    if balance == "equal":
        assert bands[0][1] == 180 and "calibration_launches" not in d["config"]
    else:
        # Scene1: sky on top, the upper band is taller; the probe is reported as the launch it is, with its cost
        assert bands[0][1] > 180 and d["config"]["calibration_launches"] == 1 and d["config"]["probe_ms"] > 0
        assert d["config"]["value_including_probe_once"] < d["value"]
