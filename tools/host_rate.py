#!/usr/bin/env python3
"""PCIe-inclusive rate: bz2hip_compress with host input/output buffers (pageable numpy memory)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bzip2_amd import Context, corpus
data = corpus.mixed(256, corpus.MIXED256_SEED)
c = Context()
c.compress(data[: 8 << 20], 9)
c.compress(data, 9)
t = time.perf_counter()
for _ in range(3):
    out = c.compress(data, 9)
dt = (time.perf_counter() - t) / 3
print(f"host-buffer compress: {data.size/dt/1e6:.0f} MB/s ({dt*1e3:.1f} ms per 256 MiB, output {len(out)} bytes)")
