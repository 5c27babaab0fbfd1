"""Sharding one .bz2 stream over several GPUs (one process per GPU).

Blocks are independent except for the bit cursor and the stream-CRC fold
(S/BZip2OutputStream.swift:94-102,131-141), so there is no data-path collective:
  1. every rank derives the same block boundaries (K1 split; serial chain, cheap),
  2. rank r compresses a contiguous range of whole blocks into a header-less bit segment
     (bz2hip_compress_blocks*),
  3. rank 0 gathers (bits, nbits, crcs) and stitches them in block order (bz2hip_stitch).
The output is byte-identical for every world size.

`engine` abstracts the compute: on a GPU it is bzip2_amd.Context (split / compress_blocks);
CPU tests inject a stand-in so the sharding and stitching logic runs under gloo.
"""
import numpy as np

from ._lib import stitch


def shard_ranges(starts, n_total, world):
    """Contiguous block ranges balanced by input bytes -> list of (byte_lo, byte_hi) per rank."""
    starts = [int(s) for s in starts]
    nb = len(starts)
    edges = starts + [int(n_total)]
    out = []
    b = 0
    for r in range(world):
        target = n_total * (r + 1) / world
        e = b
        while e < nb and edges[e + 1] <= target:
            e += 1
        if r == world - 1:
            e = nb
        out.append((edges[b] if b < nb else n_total, edges[e] if e <= nb else n_total))
        b = e
    return out


def compress_sharded(data, level, engine, rank=0, world=1, gather=None):
    """Returns the complete stream on rank 0 (None elsewhere).

    gather(obj) must return the list of every rank's obj on rank 0 (e.g. a wrapper around
    torch.distributed.gather_object); with world == 1 it may be None."""
    a = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8)
                             if not isinstance(data, np.ndarray) else data, dtype=np.uint8)
    if not 1 <= level <= 9:
        raise ValueError(f"Invalid BZip2 block size {level}")
    starts = engine.split(a, level) if a.size else []
    ranges = shard_ranges(starts, a.size, world)
    lo, hi = ranges[rank]
    if hi > lo:
        seg = engine.compress_blocks(a[lo:hi], level)
    else:
        seg = (np.zeros(0, dtype=np.uint8), 0, np.zeros(0, dtype=np.uint32))
    seg = (np.asarray(seg[0]), int(seg[1]), np.asarray(seg[2], dtype=np.uint32))
    segs = [seg] if world == 1 else gather(seg)
    if rank != 0:
        return None
    return stitch(level, segs)
