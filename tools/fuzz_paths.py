#!/usr/bin/env python3
"""Differential fuzz of the other entry paths: streaming API with ragged writes, small batch
caps (two-lane pipeline), sharded compress_blocks + stitch, stage-by-stage block pipeline.
usage: fuzz_paths.py <first_seed> <count>"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc
from bzip2_amd import Context, parallel
src = open(os.path.join(ROOT, "tools", "fuzz_gpu.py")).read().split("first, count =")[0]
ns = {"__file__": os.path.join(ROOT, "tools", "fuzz_gpu.py")}
exec(compile(src, "fuzz_gen", "exec"), ns)
gen = ns["gen"]

first, count = int(sys.argv[1]), int(sys.argv[2])
ctxs = {}
def ctx_for(batch):
    if batch not in ctxs:
        ctxs[batch] = Context(tie_rule=1, max_batch_blocks=batch)
    return ctxs[batch]

bad = 0
t0 = time.time()
for seed in range(first, first + count):
    d, level = gen(seed)
    if seed % 3 == 0:  # make it multi-block more often
        d = np.concatenate([d, d[::-1], d[: d.size // 2]])
        level = 1 + seed % 3
    rng = np.random.default_rng(seed ^ 0xABCDEF)
    want, _ = orc.compress(d, level, tie_rule=1)
    mode = seed % 4
    if mode == 0:      # streaming, ragged writes
        c = ctx_for(0)
        c.stream_begin(level)
        got = b""
        pos = 0
        while pos < d.size:
            k = int(rng.integers(1, 300000))
            c.stream_write(d[pos:pos + k].tobytes())
            pos += k
            if rng.random() < 0.5:
                got += c.stream_read()
        c.stream_finish()
        while True:
            part = c.stream_read()
            if not part:
                break
            got += part
    elif mode == 1:    # tiny batch caps: many batches, both lanes
        got = ctx_for(int(rng.integers(1, 5))).compress(d, level)
    elif mode == 2:    # sharded over 1..4 "ranks" run one after the other, the chain hop in a dict
        from bzip2_amd._lib import stitch
        world = int(rng.integers(1, 5))
        c = ctx_for(0)
        hop, segs = {}, []
        for r in range(world):
            eng = parallel.GpuShardEngine(c, 0)
            eng.bind(d)
            segs.append(parallel.run_rank(eng, level, r, world, d.size, lambda r=r: hop[r],
                                          lambda v, r=r: hop.__setitem__(r + 1, v),
                                          lookahead=int(rng.integers(1000, 200000))))
        got = stitch(level, segs)
    else:              # stage by stage: split, then whole-block compress of every block
        c = ctx_for(0)
        starts, rles, present, crc = c.rle1_split(d, level)
        ostarts = orc.split(d, level)
        ok = np.array_equal(starts, ostarts)
        for b in range(len(ostarts)):
            used, blk, pres, cr = orc.rle1_block(d[int(ostarts[b]):], level * 100000)
            ok = ok and np.array_equal(blk, rles[b]) and cr == crc[b] and np.array_equal(pres, present[b])
        got = want if ok else b"stage mismatch"
    if got != want:
        bad += 1
        print(f"MISMATCH seed={seed} mode={mode} n={d.size} level={level}", flush=True)
    if (seed - first) % 50 == 49:
        print(f"{seed - first + 1} done, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz_paths: {count} inputs, {bad} mismatches")
sys.exit(1 if bad else 0)
