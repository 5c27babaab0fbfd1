#!/usr/bin/env python3
"""Differential fuzz: random structured inputs, GPU stream vs the CPU oracle, random levels.
usage: fuzz_gpu.py <first_seed> <count>"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc
from bzip2_amd import Context, corpus

def gen(seed):
    rng = np.random.default_rng(seed)
    parts = []
    total = int(rng.integers(1, 700000))
    while sum(p.size for p in parts) < total:
        kind = int(rng.integers(0, 10))
        n = int(rng.integers(1, 200000))
        if kind == 0:
            parts.append(corpus.segment("text", n, int(rng.integers(1 << 30))))
        elif kind == 1:
            parts.append(rng.integers(0, int(rng.choice([2, 3, 4, 16, 256])), n, dtype=np.uint8))
        elif kind == 2:
            parts.append(corpus.segment("runs", n, int(rng.integers(1 << 30))))
        elif kind == 3:  # periodic, any period
            p = int(rng.integers(1, min(n, 70000) + 1))
            parts.append(np.tile(rng.integers(0, int(rng.choice([2, 256])), p, dtype=np.uint8), n // p + 1)[:n])
        elif kind == 4:  # duplicate of an earlier part
            if parts:
                q = parts[int(rng.integers(len(parts)))]
                a = int(rng.integers(0, q.size)); parts.append(q[a: a + n].copy())
        elif kind == 5:
            parts.append(np.zeros(n, dtype=np.uint8) + int(rng.integers(256)))
        elif kind == 6:
            parts.append(corpus.segment("binary", n, int(rng.integers(1 << 30))))
        elif kind == 7:
            parts.append(corpus.segment("markov", n, int(rng.integers(1 << 30))))
        elif kind == 8:  # periodic with sparse mutations
            p = int(rng.integers(1, 5000))
            x = np.tile(rng.integers(0, 256, p, dtype=np.uint8), n // p + 1)[:n].copy()
            for _ in range(int(rng.integers(0, 4))):
                x[int(rng.integers(n))] ^= 1
            parts.append(x)
        else:
            parts.append(corpus.segment("dna", n, int(rng.integers(1 << 30))))
    d = np.concatenate(parts)[:total] if parts else np.zeros(1, dtype=np.uint8)
    return d, int(rng.integers(1, 10))

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
