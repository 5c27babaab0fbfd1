"""N > 1 path on CPU: world_size-2 gloo run of the sharding + ordered stitching logic.
The compute engine is a stand-in built on the oracle (tests only); on GPUs the engine is
bzip2_amd.Context."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    def __init__(self, orc):
        self.orc = orc

    def split(self, a, level):
        return self.orc.split(a, level)

    def compress_blocks(self, a, level):
        from bzip2_amd._lib import stitch  # noqa: F401
        segs_bits, crcs = [], []
        pos = 0
        # bit-abut the blocks of this shard
        acc = bytearray()
        nb_total = 0
        import numpy as np
        bits_all = []
        while pos < a.size:
            used, blk, pres, crc = self.orc.rle1_block(a[pos:], level * 100000)
            pos += used
            bits, nbits = self.orc.block_compress(blk, pres, crc)
            bits_all.append(np.unpackbits(bits)[:nbits])
            crcs.append(crc)
        allbits = np.concatenate(bits_all) if bits_all else np.zeros(0, dtype=np.uint8)
        return np.packbits(allbits), int(allbits.size), np.array(crcs, dtype=np.uint32)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    from bzip2_amd import corpus, parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = corpus.mixed(5, 404, seg_bytes=260000)

    def gather(obj):
        out = [None] * world if rank == 0 else None
        dist.gather_object(obj, out, dst=0)
        return out

    res = parallel.compress_sharded(data, 2, OracleEngine(orc), rank, world, gather)
    if rank == 0:
        ref, _ = orc.compress(data, 2)
        single = parallel.compress_sharded(data, 2, OracleEngine(orc), 0, 1)
        q.put((res == ref, single == ref, len(ref)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_stream_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok_multi, ok_single, n = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok_multi and ok_single and n > 0


def test_shard_ranges():
    from bzip2_amd.parallel import shard_ranges
    starts = [0, 100, 200, 300, 400]
    for world in (1, 2, 3, 4, 8):
        r = shard_ranges(starts, 450, world)
        assert len(r) == world
        assert r[0][0] == 0 and r[-1][1] == 450
        for a, b in zip(r, r[1:]):
            assert a[1] == b[0]
        for lo, hi in r:
            assert lo in starts + [450] and hi in starts + [450]
    assert shard_ranges([], 0, 2) == [(0, 0), (0, 0)]
