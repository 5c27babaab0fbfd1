#!/usr/bin/env python3
"""Where the time of the incremental API goes: writes, finish, final reads (tools/stream_rate.py's loop)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bzip2_amd import Context, corpus
data = corpus.mixed(256, corpus.MIXED256_SEED)
c = Context(device=0)
piece = 1 << 20
for rep in range(6):
    t0 = time.perf_counter()
    c.stream_begin(9)
    out = []
    tr = 0.0
    for o in range(0, data.size, piece):
        c.stream_write(data[o:o + piece])
        ta = time.perf_counter()
        while c.stream_pending():
            out.append(c.stream_read(1 << 24))
        tr += time.perf_counter() - ta
    t1 = time.perf_counter()
    c.stream_finish()
    t2 = time.perf_counter()
    while c.stream_pending():
        out.append(c.stream_read(1 << 24))
    t3 = time.perf_counter()
    print(f"rep {rep}: writes+reads {1e3*(t1-t0):.1f} ms (reads inside {1e3*tr:.1f}), finish {1e3*(t2-t1):.1f}, last reads {1e3*(t3-t2):.1f}, total {1e3*(t3-t0):.1f}")
