#!/usr/bin/env python3
"""Throughput of the incremental API (bz2hip_stream_begin / write / read / finish = BZip2OutputStream.write
into a sink), host buffers both sides: 256 MiB of `mixed256` written in 1 MiB pieces, output read after
every write.  usage: python tools/stream_rate.py [piece MiB] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bzip2_amd import Context, corpus

piece = (int(sys.argv[1]) if len(sys.argv) > 1 else 1) << 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
data = corpus.mixed(256, corpus.MIXED256_SEED)
c = Context(device=0)
whole = c.compress(data, 9)
best = None
times = []
for rep in range(reps + 1):
    t0 = time.perf_counter()
    c.stream_begin(9)
    out = []
    for o in range(0, data.size, piece):
        c.stream_write(data[o:o + piece])
        while c.stream_pending():
            out.append(c.stream_read(1 << 24))
    c.stream_finish()
    while c.stream_pending():
        out.append(c.stream_read(1 << 24))
    dt = time.perf_counter() - t0
    if rep:
        times.append(dt)
        if best is None or dt < best:
            best = dt
got = b"".join(out)
print(f"stream API, {piece >> 20} MiB writes: {data.size / best / 1e6:.1f} MB/s ({best * 1e3:.1f} ms), same bytes: {got == whole}; "
      f"all reps (ms): {[round(t * 1e3, 1) for t in times]}")
