"""The fitting tool of the multi-GPU balance probe runs on the committed closed-loop iterations (profiles/r03/band_fit/) and
reproduces weights of the order the library ships (ProbeWeights, csrc/srt_capi.hip); tools/dispatches.py picks bench.py's
timed launches by position."""
import json
import os
import re
import subprocess
import sys

from conftest import ROOT


def test_band_fit_runs_on_the_committed_iterations():
    files = sorted(os.path.join(ROOT, "profiles", "r03", "band_fit", f) for f in os.listdir(os.path.join(ROOT, "profiles", "r03", "band_fit")))
    files = [f for f in files if int(f.split('iter')[1][0]) <= 4]  # the four joint iterations (5 and 6: config 5 alone, mesh weights only)
    assert len(files) == 8
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "band_fit.py")] + files, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    w = json.loads(re.search(r"weights: (\{.*\})", r.stdout).group(1))
    assert set(w) == {"groups", "node_rounds", "leaf_trips", "mesh_phases", "waves", "untraced_waves"}
    assert 300 < w["groups"] < 1500 and 500 < w["leaf_trips"] < 4000 and 2000 < w["waves"] < 12000  # the shipped 660 / 1915 / 5830 lie inside
    assert r.stdout.count("cost/time of the bands") == 24  # 8 files x N = 2, 4, 8


def test_probe_weights_in_the_library_match_the_fit_notes():
    src = open(os.path.join(ROOT, "software-raytracer_amd", "csrc", "srt_capi.hip")).read()
    notes = open(os.path.join(ROOT, "profiles", "r03", "band_fit_fit.txt")).read()
    for name, key in (("group", "groups"), ("leaf_trip", "leaf_trips"), ("wave", "waves")):
        v = float(re.search(r"\b%s = ([0-9.]+)" % name, src).group(1))
        assert '"%s": %d' % (key, round(v)) in notes, (name, v)
