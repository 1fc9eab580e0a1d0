#!/usr/bin/env python3
"""Generates tests/golden/json_reference.json: inputs + what the REFERENCE's own JSON library
(/root/reference/Raytracer/json.hpp, run through oracle/_ref/ref_json — `make -C oracle ref`) makes of them.
Data only: documents, numbers and the reference's outputs.  Needs /root/reference; the fixture travels."""
import base64
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import json_corpus  # noqa: E402

REF = os.path.join(HERE, "..", "..", "oracle", "_ref", "ref_json")


def ref_lines(docs, indent):
    r = subprocess.run([REF, "lines", str(indent)], input=b"\n".join(docs) + b"\n", capture_output=True, check=True)
    out = r.stdout.split(b"\n")[:-1]
    assert len(out) == len(docs)
    return [None if o == b"EXCEPTION" else o.replace(b"\x1e", b"\n") for o in out]


def ref_numbers(vals):
    r = subprocess.run([REF, "numbers"], input=("\n".join(v.hex() for v in vals) + "\n").encode(), capture_output=True, check=True)
    out = r.stdout.decode().split("\n")[:-1]
    assert len(out) == len(vals)
    return out


def main():
    docs = json_corpus.documents(600)
    d4, dm1 = ref_lines(docs, 4), ref_lines(docs, -1)
    nums = json_corpus.numbers(4000)
    b64 = lambda b: None if b is None else base64.b64encode(b).decode()
    fixture = {
        "generator": "tests/golden/make_json_golden.py", "reference": "Raytracer/json.hpp (nlohmann/json 3.11.2) via oracle/ref_json_harness.cpp",
        "documents": [{"in": b64(d), "dump4": b64(a), "dump": b64(b)} for d, a, b in zip(docs, d4, dm1)],
        "numbers": [[v.hex(), s] for v, s in zip(nums, ref_numbers(nums))],
    }
    with open(os.path.join(HERE, "json_reference.json"), "w") as f:
        json.dump(fixture, f, separators=(",", ":"))
    print("documents", len(docs), "rejected", sum(a is None for a in d4), "numbers", len(nums))


if __name__ == "__main__":
    main()
