#!/usr/bin/env python3
"""Compress mixed256 once, then decode it 3 times on the GPU (for rocprofv3 kernel stats of the
decoder: only the decoder's kernels start with kd_)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bzip2_amd import Context, corpus

seg = int(sys.argv[1]) if len(sys.argv) > 1 else 256
data = corpus.mixed(seg, corpus.MIXED256_SEED)
c = Context(device=0)
d_in = torch.from_numpy(data).cuda()
cap = c.L.bz2hip_compress_bound(data.size)
d_out = torch.empty(cap + 64, dtype=torch.uint8, device="cuda")
n = c.compress_device(d_in.data_ptr(), data.size, 9, d_out.data_ptr(), cap)
d_back = torch.empty(data.size + 64, dtype=torch.uint8, device="cuda")
for k in range(3):
    torch.cuda.synchronize()
    t = time.perf_counter()
    rc, dn = c.decompress_device(d_out.data_ptr(), n, d_back.data_ptr(), data.size)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"decode {k}: rc {rc} bytes {dn} {dt * 1e3:.1f} ms {data.size / dt / 1e6:.0f} MB/s ok {bool(torch.equal(d_back[:dn], d_in))}", flush=True)
