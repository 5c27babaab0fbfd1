"""Differential-fuzz building blocks shared by the campaign scripts (tools/fuzz_*.py, run by hand on
the GPU box for hours) and by the fixed-seed slice inside `pytest -m gpu` (tests/test_gpu_fuzz.py).
Generators of structured inputs, one encoder case through a chosen entry path, one decoder case.
The oracle is the checker in every case; nothing here is product code."""
import bz2

import numpy as np

from bzip2_amd import corpus

# oracle decoder code -> C ABI code
MAP = {0: 0, -1: -6, -2: -7, -3: -8, -4: -9, -5: -10, -6: -11, -7: -12, -8: -13, -9: -14, -10: -15, -11: -4}


def gen(seed, max_total=700000, max_part=200000):
    """-> (input bytes as uint8 array, level): parts of random structure -- text, tiny alphabets, runs of
    every length class, periodic stretches of any period, copies of earlier parts, constant bytes,
    binary records, Markov chains, periodic data with sparse mutations, DNA."""
    rng = np.random.default_rng(seed)
    parts = []
    total = int(rng.integers(1, max_total))
    while sum(p.size for p in parts) < total:
        kind = int(rng.integers(0, 10))
        n = int(rng.integers(1, max_part))
        if kind == 0:
            parts.append(corpus.segment("text", n, int(rng.integers(1 << 30))))
        elif kind == 1:
            parts.append(rng.integers(0, int(rng.choice([2, 3, 4, 16, 256])), n, dtype=np.uint8))
        elif kind == 2:
            parts.append(corpus.segment("runs", n, int(rng.integers(1 << 30))))
        elif kind == 3:  # periodic, any period
            p = int(rng.integers(1, min(n, 70000) + 1))
            parts.append(np.tile(rng.integers(0, int(rng.choice([2, 256])), p, dtype=np.uint8), n // p + 1)[:n])
        elif kind == 4:  # duplicate of an earlier part
            if parts:
                q = parts[int(rng.integers(len(parts)))]
                a = int(rng.integers(0, q.size))
                parts.append(q[a: a + n].copy())
        elif kind == 5:
            parts.append(np.zeros(n, dtype=np.uint8) + int(rng.integers(256)))
        elif kind == 6:
            parts.append(corpus.segment("binary", n, int(rng.integers(1 << 30))))
        elif kind == 7:
            parts.append(corpus.segment("markov", n, int(rng.integers(1 << 30))))
        elif kind == 8:  # periodic with sparse mutations
            p = int(rng.integers(1, 5000))
            x = np.tile(rng.integers(0, 256, p, dtype=np.uint8), n // p + 1)[:n].copy()
            for _ in range(int(rng.integers(0, 4))):
                x[int(rng.integers(n))] ^= 1
            parts.append(x)
        else:
            parts.append(corpus.segment("dna", n, int(rng.integers(1 << 30))))
    d = np.concatenate(parts)[:total] if parts else np.zeros(1, dtype=np.uint8)
    return d, int(rng.integers(1, 10))


def path_case(seed, orc, ctx_for, tie_rule=1, **gen_kw):
    """One input through one of the other entry paths (seed % 4): 0 streaming API with ragged writes and
    interleaved reads, 1 tiny batch caps (many batches, both lanes), 2 one to four shards + host
    stitch, 3 stage by stage (split / RLE1 / CRC).  ctx_for(batch) -> a Context(tie_rule=tie_rule,
    max_batch_blocks=batch).  -> (ok, description)"""
    from bzip2_amd import parallel
    from bzip2_amd._lib import stitch
    d, level = gen(seed, **gen_kw)
    if seed % 3 == 0:  # make it multi-block more often
        d = np.concatenate([d, d[::-1], d[: d.size // 2]])
        level = 1 + seed % 3
    rng = np.random.default_rng(seed ^ 0xABCDEF)
    want, _ = orc.compress(d, level, tie_rule=tie_rule)
    mode = seed % 4
    if mode == 0:
        c = ctx_for(0)
        c.stream_begin(level)
        got = b""
        pos = 0
        while pos < d.size:
            k = int(rng.integers(1, 300000))
            c.stream_write(d[pos:pos + k].tobytes())
            pos += k
            if rng.random() < 0.5:
                got += c.stream_read()
        c.stream_finish()
        while True:
            part = c.stream_read()
            if not part:
                break
            got += part
    elif mode == 1:
        got = ctx_for(int(rng.integers(1, 5))).compress(d, level)
    elif mode == 2:
        world = int(rng.integers(1, 5))
        c = ctx_for(0)
        hop, segs = {}, []
        for r in range(world):
            eng = parallel.GpuShardEngine(c, 0)
            eng.bind(d)
            segs.append(parallel.run_rank(eng, level, r, world, d.size, lambda r=r: hop[r],
                                          lambda v, r=r: hop.__setitem__(r + 1, v),
                                          lookahead=int(rng.integers(1000, 200000))))
        got = stitch(level, segs)
    else:
        c = ctx_for(0)
        starts, rles, present, crc = c.rle1_split(d, level)
        ostarts = orc.split(d, level)
        ok = np.array_equal(starts, ostarts)
        for b in range(len(ostarts)):
            used, blk, pres, cr = orc.rle1_block(d[int(ostarts[b]):], level * 100000)
            ok = ok and np.array_equal(blk, rles[b]) and cr == crc[b] and np.array_equal(pres, present[b])
        got = want if ok else b"stage mismatch"
    return got == want, f"seed={seed} mode={mode} n={d.size} level={level}"


def decode_case(seed, orc, ctx, **gen_kw):
    """One stream through the GPU decoder: encoded by libbz2 (odd seeds) or by the GPU encoder; every
    fourth stream is damaged first (1-3 bit flips, or a truncation) and the error code and the
    delivered prefix must equal the decoder oracle's.  -> (ok, description)"""
    d, level = gen(seed, **gen_kw)
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
    return ok, f"seed={seed} n={len(data)} level={level} rc={rc}"
