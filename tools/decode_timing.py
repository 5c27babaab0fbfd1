#!/usr/bin/env python3
"""Decode throughput (device-resident stream in, bytes out) on the bench corpus."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bzip2_amd import Context, corpus
seg = int(sys.argv[1]) if len(sys.argv) > 1 else 256
level = int(sys.argv[2]) if len(sys.argv) > 2 else 9
data = corpus.mixed(seg, corpus.MIXED256_SEED)
c = Context(device=0)
d_in = torch.from_numpy(data).cuda()
cap = c.L.bz2hip_compress_bound(data.size)
d_z = torch.empty(cap + 64, dtype=torch.uint8, device="cuda")
zl = c.compress_device(d_in.data_ptr(), data.size, level, d_z.data_ptr(), cap)
d_out = torch.empty(data.size + 64, dtype=torch.uint8, device="cuda")
for it in range(3):
    torch.cuda.synchronize()
    t = time.perf_counter()
    rc, n = c.decompress_device(d_z.data_ptr(), zl, d_out.data_ptr(), data.size)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    ok = rc == 0 and n == data.size and bool(torch.equal(d_out[:n], d_in))
    print(f"decode {seg} MiB -{level}: {data.size / dt / 1e6:8.1f} MB/s  {dt * 1e3:7.1f} ms  rc={rc} ok={ok}", flush=True)
