#!/usr/bin/env python3
"""What `bench.py --gpus 8` asks of the library, on ONE GPU and in one process (the box has one
card; eight ranks on it are not allowed): the 2 GiB stream of the `stream1g` generator (8 x 256 MiB)
  * compressed whole by bz2hip_compress_device (the check rank 0 makes after the timed region),
  * as 8 shards with the bench's own slices (nominal range + 8 MiB resident, 2 MiB lookahead),
    one "rank" after the other, segments concatenated in HBM by bz2hip_stitch_device,
  * decoded again by the GPU decoder (stream > 2^32 bits, output 2^31 bytes).
Prints one JSON line.  usage: python tools/rehearse_n8_one_gpu.py [world] [segments per rank]
"""
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bzip2_amd import Context, corpus, parallel  # noqa: E402


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    seg = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    level = 9
    dev = torch.device("cuda", 0)
    ctx = Context(device=0)
    L = ctx.L
    n_total = world * seg * corpus.MIB
    t0 = time.time()
    whole = corpus.mixed(world * seg, corpus.STREAM1G_SEED)
    print(f"[rehearse] corpus {n_total >> 20} MiB in {time.time() - t0:.1f} s", file=sys.stderr, flush=True)
    d_whole = torch.from_numpy(whole).to(dev)
    cap = L.bz2hip_compress_bound(n_total)
    d_one = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    t0 = time.time()
    n1 = ctx.compress_device(d_whole.data_ptr(), n_total, level, d_one.data_ptr(), cap)
    t_whole = time.time() - t0
    h1 = hashlib.sha256(d_one[:n1].cpu().numpy().tobytes()).hexdigest()
    print(f"[rehearse] whole stream {n1} bytes in {t_whole * 1e3:.1f} ms", file=sys.stderr, flush=True)

    hop = {}
    segs, keep = [], []
    look_segs, held_segs = 2, 8
    for r in range(world):
        lo, hi = parallel.nominal_ranges(n_total, world)[r]
        s0, s1 = lo // corpus.MIB, min(world * seg, hi // corpus.MIB + held_segs)
        end = s1 * corpus.MIB
        d_slice = d_whole[lo:end]  # the bench builds the same bytes segment by segment
        eng = parallel.GpuShardEngine(ctx, 0)
        eng.bind_device(d_slice, lo, end)
        nb = parallel.plan_rank(eng, level, r, world, n_total, lambda r=r: hop[r],
                                lambda v, r=r: hop.__setitem__(r + 1, v),
                                lookahead=min(end - hi, look_segs * corpus.MIB) if r + 1 < world else 0)
        if nb:
            d, nbits, crcs = eng.compress_device()
            d = d[: (nbits + 31) // 32 * 4 + 8].clone()
        else:
            d, nbits, crcs = torch.zeros(8, dtype=torch.uint8, device=dev), 0, np.zeros(0, dtype=np.uint32)
        keep.append(d)
        segs.append((d.data_ptr(), nbits, crcs))
        print(f"[rehearse] rank {r}: {nb} blocks, {nbits} bits", file=sys.stderr, flush=True)
    d_out = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
    n8 = ctx.stitch_device(level, segs, d_out.data_ptr(), cap)
    h8 = hashlib.sha256(d_out[:n8].cpu().numpy().tobytes()).hexdigest()

    d_back = torch.empty(n_total + 64, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    t0 = time.time()
    rc, dn = ctx.decompress_device(d_out.data_ptr(), n8, d_back.data_ptr(), n_total)
    torch.cuda.synchronize()
    t_dec = time.time() - t0
    ok = bool(rc == 0 and dn == n_total and torch.equal(d_back[:dn], d_whole))
    print(json.dumps({"world": world, "stream_bytes": n_total, "compressed_bytes": n8,
                      "total_bits": int(sum(s[1] for s in segs)),
                      "sharded_equals_whole": bool(n1 == n8 and h1 == h8), "sha256": h8,
                      "decode_rc": rc, "decode_ok": ok,
                      "whole_ms": round(t_whole * 1e3, 1), "decode_ms": round(t_dec * 1e3, 1)}))
    ctx.close()
    sys.exit(0 if (ok and h1 == h8) else 1)


if __name__ == "__main__":
    main()
