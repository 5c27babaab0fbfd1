"""A fixed-seed slice of the differential-fuzz campaigns (tools/fuzz_gpu.py, fuzz_paths.py,
fuzz_decode.py; generators in tests/fuzzgen.py) inside `pytest -m gpu`, so that every driver-run GPU
record carries differential evidence against the oracle: 400 generated inputs through
bz2hip_compress at random levels (tie rule = the reference's sorter), 100 through the other entry
paths (streaming with ragged writes, tiny batches on both lanes, 1-4 shards + stitch, stage by
stage), 200 streams through the decoder, a quarter of them damaged.  Seeds are the ranges below."""
import numpy as np
import pytest

import fuzzgen

pytestmark = pytest.mark.gpu

COMPRESS_SEEDS = range(910_000, 910_400)
PATH_SEEDS = range(920_000, 920_100)
DECODE_SEEDS = range(930_000, 930_200)
SIZES = dict(max_total=500_000, max_part=150_000)   # the campaigns use 700 000 / 200 000


def test_fuzz_slice_compress(ctx, orc):
    bad = []
    periodic = 0
    for seed in COMPRESS_SEEDS:
        d, level = fuzzgen.gen(seed, **SIZES)
        want, _ = orc.compress(d, level)
        got = ctx.compress(d, level)
        periodic += ctx.last_stats()["periodic_blocks"]
        if got != want:
            bad.append((seed, d.size, level))
    assert not bad, bad


def test_fuzz_slice_entry_paths(orc):
    from bzip2_amd import Context
    ctxs = {}

    def ctx_for(batch):
        if batch not in ctxs:
            ctxs[batch] = Context(tie_rule=1, max_batch_blocks=batch)
        return ctxs[batch]

    try:
        bad = []
        for seed in PATH_SEEDS:
            ok, what = fuzzgen.path_case(seed, orc, ctx_for, tie_rule=1, **SIZES)
            if not ok:
                bad.append(what)
        assert not bad, bad
    finally:
        for c in ctxs.values():
            c.close()


def test_fuzz_slice_decoder(ctx, orc):
    bad = []
    for seed in DECODE_SEEDS:
        ok, what = fuzzgen.decode_case(seed, orc, ctx, **SIZES)
        if not ok:
            bad.append(what)
    assert not bad, bad
