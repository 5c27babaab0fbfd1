#!/usr/bin/env python3
"""Differential fuzz of the GPU decoder: generated inputs (tools/fuzz_gpu.py's generators) encoded by
libbz2 and by the GPU encoder, decoded on the GPU and compared with the input; every fourth stream
is damaged first (1-3 bit flips or a truncation) and code + delivered prefix are compared with the
decoder oracle.   usage: fuzz_decode.py <first_seed> <count>"""
import bz2, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc
from bzip2_amd import Context
src = open(os.path.join(ROOT, "tools", "fuzz_gpu.py")).read().split("first, count =")[0]
ns = {"__file__": os.path.join(ROOT, "tools", "fuzz_gpu.py")}
exec(compile(src, "fuzz_gen", "exec"), ns)
gen = ns["gen"]
MAP = {0: 0, -1: -6, -2: -7, -3: -8, -4: -9, -5: -10, -6: -11, -7: -12, -8: -13, -9: -14, -10: -15, -11: -4}

first, count = int(sys.argv[1]), int(sys.argv[2])
ctx = Context(tie_rule=1)
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    d, level = gen(seed)
    data = d.tobytes()
    rng = np.random.default_rng(seed ^ 0x5A5A)
    stream = bz2.compress(data, level) if seed & 1 else ctx.compress(d, level)
    if seed % 4 == 3 and len(stream) > 20:
        z = bytearray(stream)
        if rng.random() < 0.3:
            z = z[: int(rng.integers(1, len(z)))]
        else:
            for _ in range(int(rng.integers(1, 4))):
                z[int(rng.integers(0, len(z)))] ^= 1 << int(rng.integers(0, 8))
        z = bytes(z)
        cap = max(len(data) * 3, 1 << 16)
        orc_rc, orc_out, _ = orc.decompress(z, cap)
        rc, out = ctx.decompress(z, cap)
        ok = (rc == MAP[orc_rc] and out == orc_out) or (orc_rc == orc.D_CAPACITY and rc == -4)
    else:
        rc, out = ctx.decompress(stream, len(data))
        ok = rc == 0 and out == data
    if not ok:
        bad += 1
        print(f"MISMATCH seed={seed} n={len(data)} level={level} rc={rc}", flush=True)
    if (seed - first) % 100 == 99:
        print(f"{seed - first + 1} done, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz_decode: {count} streams, {bad} mismatches")
sys.exit(1 if bad else 0)
