#!/usr/bin/env python3
"""Throughput on the adversarial families (SURVEY config 5), device-resident, level 9 and 1."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bzip2_amd import Context, corpus

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = mib << 20
adv = corpus.adversarial(n)
c = Context(device=0)
for name, d in adv.items():
    d_in = torch.from_numpy(np.ascontiguousarray(d)).cuda()
    cap = c.L.bz2hip_compress_bound(d.size)
    out = torch.empty(cap + 64, dtype=torch.uint8, device="cuda")
    for level in (9, 1):
        c.compress_device(d_in.data_ptr(), d.size, level, out.data_ptr(), cap)
        torch.cuda.synchronize()
        t = time.perf_counter()
        ln = c.compress_device(d_in.data_ptr(), d.size, level, out.data_ptr(), cap)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        st = c.last_stats()
        print(f"{name:12s} -{level}: {d.size/dt/1e6:9.1f} MB/s  {dt*1e3:8.1f} ms  blocks={st['blocks']:4d} "
              f"rounds/blk={st['rounds_sum']/max(st['blocks'],1):5.1f} periodic={st['periodic_blocks']} out={ln}", flush=True)
