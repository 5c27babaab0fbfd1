#!/usr/bin/env python3
"""Generates tests/golden/golden.json from the CPU oracle (oracle/bz2_oracle.c).

The reference itself ships no fixture and cannot run here, so these vectors
pin the *restatement*: they freeze today's oracle output so that later edits to the oracle or
the HIP path are caught.  Small inputs are stored verbatim (hex) with the full expected
stream; larger ones as generator name + seed with the SHA-256 of the expected stream.
Run:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np  # noqa: E402
import orc  # noqa: E402
from bzip2_amd import corpus  # noqa: E402


def small_inputs():
    rng = np.random.default_rng(20251110)
    v = {
        "empty": b"",
        "one_byte": b"x",
        "two_bytes": b"ab",
        "banana": b"banana_bandana",
        "run4": b"aaaa",
        "run5": b"aaaaa",
        "run255": b"z" * 255,
        "run256": b"z" * 256,
        "run259_then": b"q" * 259 + b"r",
        "fb510": bytes([0xFB]) * 510,
        "abab": b"ab" * 40,
        "abc_period": b"abc" * 33,
        "all_bytes": bytes(range(256)),
        "random300": rng.integers(0, 256, 300, dtype=np.uint8).tobytes(),
        "text600": corpus.text(600, 5).tobytes(),
        "lowent": rng.integers(0, 2, 700, dtype=np.uint8).tobytes(),
        "hello": b"Hello, hello, hello world! " * 12,
        # D23: 4 tables, the seed back-off branch is taken and the Java-original / Swift-text readings
        # of it give different tables (tests/test_oracle.py::test_seed_backoff_follows_the_java_original_d23)
        "d23_backoff": corpus.text(700, 5).tobytes(),
    }
    return v


def large_inputs():
    return [
        ("text", 250000, 11), ("random", 120000, 12), ("runs", 300000, 13),
        ("periodic", 200000, 14), ("binary", 200000, 15), ("markov", 200000, 16),
        ("dna", 200000, 17), ("zeros", 1000000, 18),
    ]


def main():
    out = {"small": {}, "large": []}
    for name, data in small_inputs().items():
        ent = {"input_hex": data.hex(), "streams": {}}
        for level in (1, 9):
            s, nb = orc.compress(data, level)
            ent["streams"][str(level)] = {"hex": s.hex(), "blocks": nb}
        out["small"][name] = ent
    # the same input under the OTHER reading of D23 (the Swift text's), so that a later decision on the
    # contract has its expected stream on file (oracle built with -DORC_D23_SWIFT_TEXT=1)
    d23 = small_inputs()["d23_backoff"]
    out["d23_backoff_swift_text"] = {"input": "small/d23_backoff", "streams": {
        str(level): {"hex": orc.compress_d23_swift_text(d23, level)[0].hex()} for level in (1, 9)}}
    for kind, n, seed in large_inputs():
        data = corpus.segment(kind, n, seed)
        ent = {"kind": kind, "n": n, "seed": seed,
               "input_sha256": hashlib.sha256(data.tobytes()).hexdigest(), "streams": {}}
        for level in (1, 9):
            s, nb = orc.compress(data, level)
            ent["streams"][str(level)] = {"sha256": hashlib.sha256(s).hexdigest(), "len": len(s),
                                          "blocks": nb}
        out["large"].append(ent)
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote golden.json:", len(out["small"]), "small,", len(out["large"]), "large")


if __name__ == "__main__":
    main()
