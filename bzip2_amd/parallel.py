"""Sharding ONE .bz2 stream over several GPUs (one process, or one thread, per GPU).

Blocks are independent except for the bit cursor and the stream-CRC fold
(S/BZip2OutputStream.swift:94-102,131-141), so there is no reduction between GPUs.  What is serial
is the block *chain*: block k+1 starts where block k ends (S/BZip2BlockCompressor.swift:198-227).
  1. rank r holds the nominal byte range [lo_r, hi_r) of the stream plus some lookahead (its last
     block may end beyond hi_r) and runs the run-length scan over it at once (bz2hip_shard_scan:
     the scan does not depend on where blocks start),
  2. the chain hops from rank to rank: rank r receives the start of its first block (one integer)
     from rank r-1, walks its own part of the chain (bz2hip_shard_chain: a few dependent loads per
     block) and passes the start of the next block on to rank r+1,
  3. rank r compresses its blocks into a header-less bit segment (bz2hip_shard_compress),
  4. rank 0 gathers (bits, nbits, crcs) and stitches them in block order (bz2hip_stitch on the
     host, bz2hip_stitch_device in HBM).
The output is byte-identical for every world size.

`engine` abstracts the compute: on a GPU it is GpuShardEngine (the C ABI); CPU tests inject a
stand-in so that the chain hopping, the lookahead logic and the stitch run under gloo.
"""
import queue
import threading

import numpy as np

from ._lib import stitch

ALIGN = 4096


def nominal_ranges(n_total, world):
    """[lo_r, hi_r) per rank: equal byte ranges, cut at multiples of 4 KiB."""
    n_total = int(n_total)
    cuts = [0]
    for r in range(1, world):
        c = (n_total * r // world) // ALIGN * ALIGN
        cuts.append(max(c, cuts[-1]))
    cuts.append(n_total)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def default_lookahead(level):
    """Bytes past hi_r a rank holds at first: one run-free block and a margin.  A block of long
    runs can swallow up to ~51x its size of input; run_rank then asks for more."""
    return level * 100000 + 65536


def max_lookahead(level):
    return level * 100000 * 52 + 65536


def plan_rank(engine, level, rank, world, n_total, recv_first=None, send_next=None, lookahead=None):
    """Steps 1 and 2 of one rank's part: scan, then the rank's stretch of the block chain.
    engine.load(lo, end) makes the bytes [lo, end) of the stream resident and scans them;
    engine.chain(first_rel, limit_rel, final) -> (n_blocks, next_rel); recv_first() / send_next(v)
    carry the block chain between neighbouring ranks.  Returns the number of blocks that start in
    this rank's range.  Nothing here depends on the compression of any block, so a caller with a
    second engine can plan the next stream while this one is still being compressed (bench.py)."""
    if not 1 <= level <= 9:
        raise ValueError(f"Invalid BZip2 block size {level}")
    lo, hi = nominal_ranges(n_total, world)[rank]
    look = default_lookahead(level) if lookahead is None else int(lookahead)
    first = None
    while True:
        end = min(n_total, hi + look)
        engine.load(lo, end, level)
        if first is None:
            first = 0 if rank == 0 else int(recv_first())
        final = end == n_total
        if first >= hi or first < lo:
            # no block starts here: the previous rank's last block swallowed this range (or, with a
            # stream that is not final, ran into the end of the data)
            nblocks, nxt = 0, first
            break
        nblocks, nxt_rel = engine.chain(first - lo, hi - lo, final)
        nxt = lo + nxt_rel
        if nxt >= hi or final:
            break
        if look >= max_lookahead(level):
            raise RuntimeError("block chain did not close within the maximum lookahead")
        look = min(look * 4, max_lookahead(level))  # the last block ran into the end of the slice
    if rank + 1 < world and send_next is not None:
        send_next(nxt)
    return nblocks


def run_rank(engine, level, rank, world, n_total, recv_first=None, send_next=None, lookahead=None):
    """One rank's part of the stream: plan_rank, then engine.compress() -> the rank's segment
    (bits, nbits, crcs)."""
    nblocks = plan_rank(engine, level, rank, world, n_total, recv_first, send_next, lookahead)
    if nblocks == 0:
        return engine.empty()
    return engine.compress()


class RankPipeline:
    """Streams back to back on one rank, as three overlapping stages on alternating engines:
      plan     (thread)  scan + this rank's stretch of the block chain      -- plan(engine) -> n_blocks
      compress (thread)  engine.compress() -> segment
      ship     (caller)  hand the segment on (gather, stitch)               -- ship(engine, segment)
    While engine j % E compresses stream j, the plan of stream j + 1 runs on the next engine, so the
    serial hop of the chain from rank to rank is hidden behind compression instead of idling every
    GPU after the first; the segment of stream j is shipped while stream j + 1 is compressed.  An
    engine's segment buffer is not reused before its last segment has been shipped.  Each stage is
    used by one thread only (give each its own communicator: the plan stage the hop, the caller
    the gather).  run(k) does all the work of its k streams between its call and its return."""

    def __init__(self, engines, plan, ship, thread_init=None):
        self.engines = list(engines)
        self.plan, self.ship, self.thread_init = plan, ship, thread_init
        self.n = 0  # number of the next stream
        self._end = 0
        self._q_plan, self._q_planned, self._q_done = queue.Queue(), queue.Queue(), queue.Queue()
        self._shipped = [threading.Semaphore(1) for _ in self.engines]
        self._threads = [threading.Thread(target=self._planner, daemon=True),
                         threading.Thread(target=self._compressor, daemon=True)]
        for t in self._threads:
            t.start()

    def _planner(self):
        if self.thread_init:
            self.thread_init()
        while True:
            j = self._q_plan.get()
            if j is None:
                self._q_planned.put(None)
                return
            try:
                self._q_planned.put((j, self.plan(self.engines[j % len(self.engines)]), None))
            except BaseException as ex:  # raised again by run()
                self._q_planned.put((j, 0, ex))

    def _compressor(self):
        if self.thread_init:
            self.thread_init()
        while True:
            item = self._q_planned.get()
            if item is None:
                return
            j, nblocks, ex = item
            seg = None
            if ex is None:
                try:
                    e = j % len(self.engines)
                    self._shipped[e].acquire()  # this engine's previous segment has left its buffer
                    if j + 1 < self._end:
                        self._q_plan.put(j + 1)
                    seg = self.engines[e].compress() if nblocks else self.engines[e].empty()
                except BaseException as ex2:
                    ex = ex2
            self._q_done.put((j, seg, ex))

    def run(self, k):
        """k streams; returns ship()'s results in stream order."""
        results = []
        if k <= 0:
            return results
        self._end = self.n + k
        self._q_plan.put(self.n)
        for _ in range(k):
            j, seg, ex = self._q_done.get()
            if ex is not None:
                raise ex
            e = j % len(self.engines)
            results.append(self.ship(self.engines[e], seg))
            self._shipped[e].release()
            self.n = j + 1
        return results

    def close(self):
        self._q_plan.put(None)
        for t in self._threads:
            t.join(timeout=30)


def compress_sharded(data, level, engine, rank=0, world=1, gather=None, recv_first=None,
                     send_next=None, lookahead=None):
    """Returns the complete stream on rank 0 (None elsewhere).  `data` is the whole stream as a host
    array (every rank slices its own part from it); gather(obj) must return the list of every
    rank's obj on rank 0 (e.g. a wrapper around torch.distributed.gather_object); with world == 1
    it may be None."""
    a = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8)
                             if not isinstance(data, np.ndarray) else data, dtype=np.uint8)
    engine.bind(a)
    seg = run_rank(engine, level, rank, world, a.size, recv_first, send_next, lookahead)
    seg = (np.asarray(seg[0]), int(seg[1]), np.asarray(seg[2], dtype=np.uint32))
    segs = [seg] if world == 1 else gather(seg)
    if rank != 0:
        return None
    return stitch(level, segs)


class GpuShardEngine:
    """The C ABI (bzip2_amd.Context) as the engine of run_rank; slices live in HBM as torch tensors
    (torch is used for device memory only)."""

    def __init__(self, ctx, device=0):
        import torch
        self.torch = torch
        self.ctx = ctx
        self.dev = torch.device("cuda", device)
        self.host = None
        self.d_in = None
        self.loaded = (None, None)
        self.nblocks = 0
        self.d_out = None
        self.level = 0

    def bind(self, host_array):
        self.host = host_array
        self.loaded = (None, None)

    def bind_device(self, d_slice, lo, end):
        """slice [lo, end) of the stream already resident in HBM (bench: no host copy)"""
        self.host = None
        self.d_in = d_slice
        self.resident = (lo, end)
        self.loaded = (None, None)

    def load(self, lo, end, level):
        self.level = level
        if self.host is not None:
            if self.loaded != (lo, end):
                self.d_in = self.torch.from_numpy(self.host[lo:end]).to(self.dev) if end > lo else \
                    self.torch.zeros(4, dtype=self.torch.uint8, device=self.dev)
                self.loaded = (lo, end)
            off = 0
        else:
            rlo, rend = self.resident
            if lo < rlo or end > rend:
                raise RuntimeError(f"rank holds [{rlo}, {rend}) of the stream, the chain needs [{lo}, {end})")
            off = lo - rlo
        self.base_ptr = self.d_in.data_ptr() + off
        self.avail = end - lo
        self.ctx.shard_scan(self.base_ptr, self.avail, level)

    def chain(self, first_rel, limit_rel, final):
        self.nblocks, nxt = self.ctx.shard_chain(first_rel, limit_rel, final)
        self.in_bytes = nxt - first_rel
        return self.nblocks, nxt

    def compress_device(self):
        """-> (device tensor, nbits, crcs)"""
        cap = self.ctx.L.bz2hip_compress_bound(self.in_bytes)
        if self.d_out is None or self.d_out.numel() < cap + 64:
            self.d_out = self.torch.empty(cap + 64, dtype=self.torch.uint8, device=self.dev)
        nbits, crcs = self.ctx.shard_compress(self.d_out.data_ptr(), cap, self.nblocks + 1)
        return self.d_out, nbits, crcs

    def compress(self):
        d, nbits, crcs = self.compress_device()
        return d[: (nbits + 7) // 8].cpu().numpy(), nbits, crcs

    def empty(self):
        return np.zeros(0, dtype=np.uint8), 0, np.zeros(0, dtype=np.uint32)
