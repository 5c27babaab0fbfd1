"""The C-ABI library loads and exports every symbol include/bz2hip.h declares (no compute)."""
import os
import re

import numpy as np

import bzip2_amd
from bzip2_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "bz2hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bz2hip_[a-z0-9_]+)\s*\(", src)))


def test_exports_match_header():
    L = bzip2_amd.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"libbz2hip.so lacks {n}"
    assert sorted(_lib.SYMBOLS) == names


def test_version_and_bound():
    L = bzip2_amd.load()
    assert b"gfx950" in L.bz2hip_version()
    assert L.bz2hip_compress_bound(0) >= 14
    assert L.bz2hip_compress_bound(1 << 20) > (1 << 20)


def test_stitch_host_only(orc):
    # bz2hip_stitch is pure host code: header + blocks + CRC fold + trailer
    from bzip2_amd import corpus
    data = corpus.mixed(2, 31, seg_bytes=150000)
    segs = []
    pos = 0
    while pos < data.size:
        used, blk, pres, crc = orc.rle1_block(data[pos:], 100000)
        pos += used
        bits, nbits = orc.block_compress(blk, pres, crc, 100000)
        segs.append((bits, nbits, np.array([crc], dtype=np.uint32)))
    assert _lib.stitch(1, segs) == orc.compress(data, 1)[0]
    assert _lib.stitch(9, []) == bytes.fromhex("425a683917724538509000000000")


def test_product_does_not_import_oracle():
    # the product package must not reference the oracle
    pkg = os.path.join(ROOT, "bzip2_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "import orc" not in txt and "bz2_oracle" not in txt and "libbz2oracle" not in txt, f
