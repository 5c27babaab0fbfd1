#!/usr/bin/env python3
"""Times whole-stream compression of inputs whose blocks are perfect powers (all-zero input,
"ab" repeated) under the three tie rules; shows what the reference-exact origin pointer costs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bzip2_amd import Context

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
inputs = {"zeros": np.zeros(mib << 20, dtype=np.uint8),
          "ab": np.tile(np.frombuffer(b"ab", dtype=np.uint8), (mib << 20) // 2)}
for rule, name in ((0, "reference"), (1, "first"), (2, "last")):
    c = Context(tie_rule=rule, profile=True)
    for k, d in inputs.items():
        c.compress(d[: 2 << 20], 9)
        c.profile_reset()
        t = time.perf_counter()
        out = c.compress(d, 9)
        dt = time.perf_counter() - t
        st = c.last_stats()
        p = c.profile_get()
        print(f"{name:10s} {k:6s} {mib} MiB: {dt*1e3:9.1f} ms  {d.size/dt/1e6:9.1f} MB/s  blocks={st['blocks']} "
              f"periodic={st['periodic_blocks']} tie_ms={p['bwt_tie']['ms']:.1f} out={len(out)}", flush=True)
    c.close()
