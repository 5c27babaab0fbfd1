#!/usr/bin/env python3
"""Differential fuzz of the GPU decoder: generated inputs (tests/fuzzgen.py) encoded by libbz2 and by
the GPU encoder, decoded on the GPU and compared with the input; every fourth stream is damaged
first (1-3 bit flips or a truncation) and code + delivered prefix are compared with the decoder
oracle.   usage: fuzz_decode.py <first_seed> <count>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc
from bzip2_amd import Context
from fuzzgen import decode_case

first, count = int(sys.argv[1]), int(sys.argv[2])
ctx = Context(tie_rule=1)
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    ok, what = decode_case(seed, orc, ctx)
    if not ok:
        bad += 1
        print(f"MISMATCH {what}", flush=True)
    if (seed - first) % 100 == 99:
        print(f"{seed - first + 1} done, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz_decode: {count} streams, {bad} mismatches")
sys.exit(1 if bad else 0)
