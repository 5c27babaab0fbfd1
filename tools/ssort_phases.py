#!/usr/bin/env python3
"""Experiment: phase breakdown of k2_ssort (needs a library built with -DEXP_TIMING, BZ2HIP_LIB)."""
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
c.L.bz2hip_debug_phases(buf)
hist = (C.c_ulonglong * 4)()
c.L.bz2hip_debug_hist(hist)
c.compress_device(d_in.data_ptr(), data.size, 9, out.data_ptr(), out.numel() - 64)
c.L.bz2hip_debug_phases(buf)
c.L.bz2hip_debug_hist(hist)
print("sorted elements by largest group of their window (<=8, <=32, <=128, <=1024):", list(hist))
v = list(buf)
tot = sum(v)
for name, x in zip(["load+heads", "compact", "bitonic", "classify", "emit", "pair flags", "group flags", "counts"], v):
    print(f"{name:14s} {x:14d} ticks  {100.0 * x / tot:5.1f} %")
