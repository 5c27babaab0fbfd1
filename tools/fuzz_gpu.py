#!/usr/bin/env python3
"""Differential fuzz: random structured inputs, GPU stream vs the CPU oracle, random levels.
usage: fuzz_gpu.py <first_seed> <count>      (FUZZ_TIE=0/1/2 selects the tie rule on both sides)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc
from bzip2_amd import Context
from fuzzgen import gen

if __name__ == "__main__":
    first, count = int(sys.argv[1]), int(sys.argv[2])
    TIE = int(os.environ.get("FUZZ_TIE", "0"))
    c = Context(tie_rule=TIE)
    bad = 0
    t0 = time.time()
    for seed in range(first, first + count):
        d, level = gen(seed)
        want, _ = orc.compress(d, level, tie_rule=TIE)
        got = c.compress(d, level)
        if got != want:
            bad += 1
            print(f"MISMATCH seed={seed} n={d.size} level={level} periodic={c.last_stats()['periodic_blocks']}", flush=True)
        if (seed - first) % 50 == 49:
            print(f"{seed - first + 1} done, {bad} bad, {time.time() - t0:.0f} s", flush=True)
    print(f"fuzz: {count} inputs, {bad} mismatches")
    sys.exit(1 if bad else 0)
