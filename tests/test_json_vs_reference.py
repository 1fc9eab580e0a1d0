"""Scene wire format pinned against the REFERENCE'S OWN JSON library.

The reference reads and writes scenes with its vendored nlohmann/json (Raytracer/json.hpp;
Scene.hpp:34 parse, :89-99 dump(4)).  That header is the one part of the reference that compiles in
this image as it lies, so oracle/ref_json_harness.cpp drives it directly:
  * tests/golden/json_reference.json holds 600 documents (a third byte-mutated: accept/reject
    agreement) and 4000 numbers with the reference library's outputs — checked everywhere;
  * when oracle/_ref/ref_json is present (this container; built by `make -C oracle ref`), a larger
    live differential run and the shipped scene files go through the reference library as well.
The code under test is software-raytracer_amd/host/json_min.hpp (parser, Grisu2 writer, layout)."""
import base64
import json
import os
import subprocess

import pytest

import json_corpus
from conftest import SCENE_NAMES, scene_path

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "..", "oracle", "_ref", "ref_json")
needs_ref = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/ref_json not built (needs /root/reference)")


def _b(x):
    return None if x is None else base64.b64decode(x)


def _mine(srt, doc, indent):
    t = srt.host.json_roundtrip(doc, indent)
    return None if t is None else t.encode("utf-8", "surrogateescape")


def test_golden_documents(srt):
    fx = json.load(open(os.path.join(HERE, "golden", "json_reference.json")))
    assert len(fx["documents"]) >= 500
    rejected = 0
    for d in fx["documents"]:
        doc = _b(d["in"])
        assert _mine(srt, doc, 4) == _b(d["dump4"]), doc
        assert _mine(srt, doc, -1) == _b(d["dump"]), doc
        rejected += d["dump"] is None
    assert 100 < rejected < 400  # the corpus exercises both outcomes


def test_golden_numbers(srt):
    fx = json.load(open(os.path.join(HERE, "golden", "json_reference.json")))
    assert len(fx["numbers"]) >= 4000
    for hx, want in fx["numbers"]:
        assert srt.host.format_double(float.fromhex(hx)) == want, hx


@needs_ref
def test_live_differential_documents(srt):
    docs = json_corpus.documents(6000, seed=12345)
    for indent in (4, -1):
        r = subprocess.run([REF, "lines", str(indent)], input=b"\n".join(docs) + b"\n", capture_output=True, check=True)
        ref = r.stdout.split(b"\n")[:-1]
        assert len(ref) == len(docs)
        for doc, want in zip(docs, ref):
            want = None if want == b"EXCEPTION" else want.replace(b"\x1e", b"\n")
            assert _mine(srt, doc, indent) == want, doc


@needs_ref
def test_live_differential_numbers(srt):
    vals = json_corpus.numbers(60000, seed=99)
    r = subprocess.run([REF, "numbers"], input=("\n".join(v.hex() for v in vals) + "\n").encode(), capture_output=True, check=True)
    ref = r.stdout.decode().split("\n")[:-1]
    assert len(ref) == len(vals)
    for v, want in zip(vals, ref):
        assert srt.host.format_double(v) == want, v.hex()


@needs_ref
@pytest.mark.parametrize("name", SCENE_NAMES)
def test_shipped_scenes_through_the_reference_library(srt, name):
    """file --(reference parse + dump(4))--> same bytes --(our Scene load + Dump)--> same bytes."""
    path = scene_path(name)
    raw = open(path, "rb").read()
    r = subprocess.run([REF, "dump", path], capture_output=True, check=True)
    assert r.stdout == raw
    assert srt.host.Scene(path).dump().encode() == raw
