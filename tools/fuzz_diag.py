#!/usr/bin/env python3
"""Locates the stage in which a fuzz seed differs from the oracle."""
import os, sys, bz2
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import orc
from bzip2_amd import Context
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzzgen import gen
tie = int(os.environ.get("FUZZ_TIE", "1"))
c = Context(tie_rule=tie)
for seed in map(int, sys.argv[1:]):
    d, level = gen(seed)
    want, nb = orc.compress(d, level, tie_rule=tie)
    got = c.compress(d, level)
    try:
        rt = bz2.decompress(got) == d.tobytes()
    except Exception as e:
        rt = repr(e)
    print(f"seed {seed}: n={d.size} level={level} blocks={nb} equal={got == want} len {len(got)}/{len(want)} roundtrip={rt}")
    starts, rles, present, crc = c.rle1_split(d, level)
    ostarts = orc.split(d, level)
    print("  starts equal:", np.array_equal(starts, ostarts))
    for b in range(len(ostarts)):
        used, blk, pres, cr = orc.rle1_block(d[int(ostarts[b]):], level * 100000)
        ok_rle = np.array_equal(blk, rles[b]) and cr == crc[b]
        (gbw, go, gr), = c.bwt([blk], level * 100000)
        ob, oo, ok_ = orc.bwt(blk, level * 100000, tie)
        eqb = np.array_equal(gbw, ob)
        print(f"  block {b}: n={blk.size} rle_ok={ok_rle} bwt_eq={eqb} orig {go}/{oo} rounds={gr} k={ok_}")
        if not eqb:
            diff = np.nonzero(gbw != ob)[0]
            print("    first diffs at", diff[:10], "count", diff.size)
            np.save(os.path.join(ROOT, "gpurun_out", f"fuzz_block_{seed}_{b}.npy"), blk)
