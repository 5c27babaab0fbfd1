#!/usr/bin/env python3
"""Differential fuzz of the other entry paths: streaming API with ragged writes, small batch
caps (two-lane pipeline), sharded compress_blocks + stitch, stage-by-stage block pipeline.
usage: fuzz_paths.py <first_seed> <count>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc
from bzip2_amd import Context
from fuzzgen import path_case

first, count = int(sys.argv[1]), int(sys.argv[2])
ctxs = {}
def ctx_for(batch):
    if batch not in ctxs:
        ctxs[batch] = Context(tie_rule=1, max_batch_blocks=batch)
    return ctxs[batch]

bad = 0
t0 = time.time()
for seed in range(first, first + count):
    ok, what = path_case(seed, orc, ctx_for, tie_rule=1)
    if not ok:
        bad += 1
        print(f"MISMATCH {what}", flush=True)
    if (seed - first) % 50 == 49:
        print(f"{seed - first + 1} done, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz_paths: {count} inputs, {bad} mismatches")
sys.exit(1 if bad else 0)
