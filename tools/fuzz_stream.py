#!/usr/bin/env python3
"""Differential fuzz of the asynchronous stream path against the CPU oracle: multi-megabyte inputs (long runs
included, so that blocks swallow several hand-overs), small chunks (BZ2HIP_STREAM_CHUNK_MIB=1 by default), random
write sizes, reads interleaved at random, several streams per context, now and then an abandoned stream.
usage: fuzz_stream.py <first_seed> <count>"""
import os, sys, time
os.environ.setdefault("BZ2HIP_STREAM_CHUNK_MIB", "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc
from bzip2_amd import Context, corpus


def gen(rng):
    parts = []
    total = int(rng.integers(1, 24_000_000))
    while sum(p.size for p in parts) < total:
        kind = int(rng.integers(0, 7))
        n = int(rng.integers(1, 6_000_000))
        if kind == 0:
            parts.append(corpus.segment("text", n, int(rng.integers(1 << 30))))
        elif kind == 1:
            parts.append(rng.integers(0, int(rng.choice([2, 4, 256])), n, dtype=np.uint8))
        elif kind == 2:
            parts.append(corpus.segment("runs", n, int(rng.integers(1 << 30))))
        elif kind == 3:
            parts.append(np.full(n, int(rng.integers(256)), dtype=np.uint8))  # one long run
        elif kind == 4:
            p = int(rng.integers(1, 3000))
            parts.append(np.tile(rng.integers(0, 256, p, dtype=np.uint8), n // p + 1)[:n])
        elif kind == 5:
            parts.append(corpus.segment("markov", n, int(rng.integers(1 << 30))))
        else:
            parts.append(np.repeat(rng.integers(0, 256, n // 300 + 1, dtype=np.uint8), 300)[:n])  # runs of 300
    return np.concatenate(parts)[:total], int(rng.integers(1, 10))


first, count = int(sys.argv[1]), int(sys.argv[2])
c = Context(device=0)
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    d, level = gen(rng)
    want, _ = orc.compress(d, level, threads=8)
    if rng.integers(8) == 0:  # an abandoned stream first
        c.stream_begin(int(rng.integers(1, 10)))
        c.stream_write(d[: int(rng.integers(0, d.size + 1))])
    c.stream_begin(level)
    out, pos = [], 0
    big = int(rng.choice([5000, 300_000, 5_000_000]))
    while pos < d.size:
        k = int(rng.integers(1, big))
        c.stream_write(d[pos:pos + k])
        pos += k
        if rng.integers(3) == 0:
            out.append(c.stream_read(1 << 22))
    c.stream_finish()
    while c.stream_pending():
        out.append(c.stream_read(1 << 22))
    if b"".join(out) != want:
        bad += 1
        print(f"MISMATCH seed={seed} n={d.size} level={level}", flush=True)
    if (seed - first) % 10 == 9:
        print(f"{seed - first + 1} done, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz_stream: {count} streams, {bad} mismatches")
sys.exit(1 if bad else 0)
