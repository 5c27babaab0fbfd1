#!/usr/bin/env python3
"""Experiment: where k4_lengths spends its time (needs a library built with -DEXP_TIMING, BZ2HIP_LIB)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bzip2_amd import Context, corpus
data = corpus.mixed(256, corpus.MIXED256_SEED)
d_in = torch.from_numpy(data).cuda()
c = Context(device=0)
out = torch.empty(c.L.bz2hip_compress_bound(data.size) + 64, dtype=torch.uint8, device="cuda")
buf = (C.c_ulonglong * 8)()
c.compress_device(d_in.data_ptr(), data.size, 9, out.data_ptr(), out.numel() - 64)
c.L.bz2hip_debug_k4(buf)
c.compress_device(d_in.data_ptr(), data.size, 9, out.data_ptr(), out.numel() - 64)
c.L.bz2hip_debug_k4(buf)
v = list(buf)
names = ["sum partial histograms", "rank sort", "allocate (all)", "scatter lengths", "  parent pointers", "  relocate search", "-", "-"]
# 100 MHz wall clock; 4 launches x tables
for n_, x in zip(names, v):
    print(f"{n_:24s} {x:12d} ticks")
