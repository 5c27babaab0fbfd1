"""N > 1 path on CPU: world_size-2 gloo run of the sharding logic (nominal ranges, lookahead,
block chain hopping from rank to rank, ordered stitching).  The compute engine is a stand-in built
on the oracle (tests only); on GPUs the engine is bzip2_amd.parallel.GpuShardEngine (the C ABI)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    """Same interface as GpuShardEngine, computed by the CPU oracle."""

    def __init__(self, orc):
        self.orc = orc

    def bind(self, a):
        self.a = a

    def load(self, lo, end, level):
        self.lo, self.end, self.level = lo, end, level
        self.loads = getattr(self, "loads", 0) + 1

    def chain(self, first_rel, limit_rel, final):
        sl = self.a[self.lo:self.end]
        pos = first_rel
        self.blocks = []
        while pos < limit_rel and pos < sl.size:
            used, blk, pres, crc = self.orc.rle1_block(sl[pos:], self.level * 100000)
            if pos + used == sl.size and not final:
                break  # ran into the end of the slice: not provably complete
            self.blocks.append((blk, pres, crc))
            pos += used
        return len(self.blocks), pos

    def compress(self):
        bits_all, crcs = [], []
        for blk, pres, crc in self.blocks:
            bits, nbits = self.orc.block_compress(blk, pres, crc, self.level * 100000)
            bits_all.append(np.unpackbits(bits)[:nbits])
            crcs.append(crc)
        allbits = np.concatenate(bits_all)
        return np.packbits(allbits), int(allbits.size), np.array(crcs, dtype=np.uint32)

    def empty(self):
        return np.zeros(0, dtype=np.uint8), 0, np.zeros(0, dtype=np.uint32)


def _data():
    from bzip2_amd import corpus
    # zeros at the rank boundary: the block that starts before it swallows far more than the
    # default lookahead of the small test setting
    return np.concatenate((corpus.mixed(3, 404, seg_bytes=260000), np.zeros(700000, dtype=np.uint8),
                           corpus.mixed(2, 405, seg_bytes=200000)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    from bzip2_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = _data()

    def gather(obj):
        out = [None] * world if rank == 0 else None
        dist.gather_object(obj, out, dst=0)
        return out

    def recv_first():
        t = torch.zeros(1, dtype=torch.int64)
        dist.recv(t, src=rank - 1)
        return int(t.item())

    def send_next(v):
        dist.send(torch.tensor([v], dtype=torch.int64), dst=rank + 1)

    eng = OracleEngine(orc)
    res = parallel.compress_sharded(data, 2, eng, rank, world, gather, recv_first, send_next, lookahead=50000)
    if rank == 0:
        ref, _ = orc.compress(data, 2)
        single = parallel.compress_sharded(data, 2, OracleEngine(orc), 0, 1)
        q.put((res == ref, single == ref, len(ref), eng.loads))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_stream_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok_multi, ok_single, n, loads = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok_multi and ok_single and n > 0
    assert loads > 1  # rank 0's last block ran past its lookahead: the slice was extended


def _pipe_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    from bzip2_amd import corpus, parallel
    from bzip2_amd._lib import stitch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hop_group = dist.new_group(backend="gloo")  # used by the plan thread only
    datas = [corpus.mixed(3, 77, seg_bytes=190000), np.concatenate((corpus.text(250000, 9), np.zeros(300000, dtype=np.uint8)))]
    level = 1
    engines = []
    for d in datas:
        e = OracleEngine(orc)
        e.bind(d)
        engines.append(e)

    def recv_first():
        t = torch.zeros(1, dtype=torch.int64)
        dist.recv(t, src=rank - 1, group=hop_group)
        return int(t.item())

    def send_next(v):
        dist.send(torch.tensor([v], dtype=torch.int64), dst=rank + 1, group=hop_group)

    def plan(eng):
        return parallel.plan_rank(eng, level, rank, world, eng.a.size, recv_first, send_next, lookahead=40000)

    def ship(eng, seg):
        out = [None] * world if rank == 0 else None
        dist.gather_object((np.asarray(seg[0]), int(seg[1]), np.asarray(seg[2], dtype=np.uint32)), out, dst=0)
        return stitch(level, out) if rank == 0 else None

    pipe = parallel.RankPipeline(engines, plan, ship)
    res = pipe.run(3) + pipe.run(2)  # two regions: the stream counter runs on
    pipe.close()
    if rank == 0:
        refs = [orc.compress(d, level)[0] for d in datas]
        q.put([res[j] == refs[j % 2] for j in range(5)])
    dist.barrier()
    dist.destroy_process_group()


def test_rank_pipeline_gloo_world2():
    """plan / compress / ship overlapped over consecutive streams on alternating engines: every
    stream comes out as the oracle's, in order."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_pipe_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    oks = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert oks == [True] * 5


@pytest.mark.parametrize("world", [1, 2, 3, 5, 8])
def test_chain_hopping_sequential(orc, world):
    """All ranks in one process, in rank order (the hop is a variable): every world size gives the
    oracle's stream, also when ranks end up with no block at all."""
    from bzip2_amd import parallel
    from bzip2_amd._lib import stitch
    data = _data()
    ref, _ = orc.compress(data, 3)
    hop = {}
    segs = []
    for r in range(world):
        eng = OracleEngine(orc)
        eng.bind(data)
        segs.append(parallel.run_rank(eng, 3, r, world, data.size, lambda r=r: hop[r],
                                      lambda v, r=r: hop.__setitem__(r + 1, v), lookahead=30000))
    assert stitch(3, segs) == ref
    # tiny input, more ranks than blocks
    tiny = data[:1000]
    ref, _ = orc.compress(tiny, 1)
    hop = {}
    segs = []
    for r in range(4):
        eng = OracleEngine(orc)
        eng.bind(tiny)
        segs.append(parallel.run_rank(eng, 1, r, 4, tiny.size, lambda r=r: hop[r],
                                      lambda v, r=r: hop.__setitem__(r + 1, v)))
    assert stitch(1, segs) == ref


def test_nominal_ranges():
    from bzip2_amd.parallel import nominal_ranges
    for n in (0, 1, 4095, 4096, 1 << 20, (1 << 30) + 12345):
        for world in (1, 2, 3, 4, 8):
            r = nominal_ranges(n, world)
            assert len(r) == world
            assert r[0][0] == 0 and r[-1][1] == n
            for a, b in zip(r, r[1:]):
                assert a[1] == b[0] and a[0] <= a[1]
            for lo, hi in r[:-1]:
                assert hi % 4096 == 0


def test_stitch_fast_path_random_segments():
    """bz2hip_stitch (64-bit funnel shift) against a plain numpy bit concatenation: segments of every
    length class, including empty ones and ones shorter than a byte."""
    from bzip2_amd._lib import stitch
    rng = np.random.default_rng(5)
    for trial in range(30):
        nseg = int(rng.integers(0, 9))
        segs, bits_all = [], []
        hdr = np.unpackbits(np.frombuffer(b"BZh7", dtype=np.uint8))
        scrc = 0
        for _ in range(nseg):
            nbits = int(rng.choice([0, 1, 5, 7, 8, 9, 63, 64, 65, 127, 1000, 4099, 70001]))
            raw = rng.integers(0, 256, (nbits + 7) // 8 + 8, dtype=np.uint8)  # garbage after the last bit
            crcs = rng.integers(0, 1 << 32, int(rng.integers(0, 4)), dtype=np.uint64).astype(np.uint32)
            segs.append((raw, nbits, crcs))
            bits_all.append(np.unpackbits(raw)[:nbits])
            for c in crcs:
                scrc = (((scrc << 1) | (scrc >> 31)) & 0xffffffff) ^ int(c)
        tail = np.unpackbits(np.frombuffer(bytes.fromhex("177245385090") + scrc.to_bytes(4, "big"), dtype=np.uint8))
        want = np.packbits(np.concatenate([hdr] + bits_all + [tail])).tobytes()
        assert stitch(7, segs) == want, trial
