#!/usr/bin/env python3
"""Throughput on the adversarial families (SURVEY config 5), device-resident, level 9 and 1.
For every family and level: the FIRST call on a fresh context (periodic blocks: the replays of the
reference's sorter run, one per distinct block) and a LATER call (their origin pointers come from
the context's cache), with the replay counts."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bzip2_amd import Context, corpus

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = mib << 20
adv = corpus.adversarial(n)
warm = Context(device=0)
warm.compress(corpus.text(2_000_000, 1), 9)   # workspace sizes do not count as "first call"
warm.close()
for name, d in adv.items():
    d_in = torch.from_numpy(np.ascontiguousarray(d)).cuda()
    for level in (9, 1):
        c = Context(device=0)
        cap = c.L.bz2hip_compress_bound(d.size)
        out = torch.empty(cap + 64, dtype=torch.uint8, device="cuda")
        # allocate the workspace outside the clock (a different input of the same size)
        c.compress_device(torch.from_numpy(corpus.segment("random", d.size, 3)).cuda().data_ptr(), d.size, level, out.data_ptr(), cap)
        torch.cuda.synchronize()
        t = time.perf_counter()
        c.compress_device(d_in.data_ptr(), d.size, level, out.data_ptr(), cap)
        torch.cuda.synchronize()
        first = time.perf_counter() - t
        ts1 = c.tie_stats()
        t = time.perf_counter()
        ln = c.compress_device(d_in.data_ptr(), d.size, level, out.data_ptr(), cap)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        st = c.last_stats()
        ts2 = c.tie_stats()
        print(f"{name:12s} -{level}: later {d.size/dt/1e6:9.1f} MB/s {dt*1e3:8.1f} ms | first {d.size/first/1e6:9.1f} MB/s {first*1e3:8.1f} ms "
              f"replays {ts1['replays']} | blocks={st['blocks']:4d} rounds/blk={st['rounds_sum']/max(st['blocks'],1):5.1f} "
              f"periodic={st['periodic_blocks']} later-replays={ts2['replays'] - ts1['replays']} out={ln}", flush=True)
        c.close()
