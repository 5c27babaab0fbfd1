"""Decoder restatement (oracle/bz2_decode_oracle.c) against libbz2: a decoder is
pinned by the format, so Python's bz2 is a full check of the decoded bytes; the error paths follow
S/BZip2InputStream.swift:172-178,228-241 and S/BZip2BlockDecompressor.swift:228-236,454-457."""
import bz2
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import orc  # noqa: E402
from bzip2_amd import corpus  # noqa: E402


def _inputs():
    rng = np.random.default_rng(7)
    yield "empty", b""
    yield "one", b"x"
    yield "text", corpus.text(300000, 1).tobytes()
    yield "random", rng.integers(0, 256, 250000, dtype=np.uint8).tobytes()
    yield "runs", (b"a" * 1000 + b"b" * 4 + b"c" * 5 + b"d" * 259 + b"e" * 3) * 300
    yield "zeros", bytes(1 << 20)
    yield "ff255", b"\xff" * 255 * 40
    yield "mixed", corpus.mixed(2, 99).tobytes()
    yield "count_eq_byte", (b"\x05" * 9 + b"\x00" * 4 + b"\x04" * 8) * 1000


@pytest.mark.parametrize("name,data", list(_inputs()), ids=[n for n, _ in _inputs()])
@pytest.mark.parametrize("level", [1, 9])
def test_decode_streams_from_both_encoders(name, data, level):
    for stream in (bz2.compress(data, level), orc.compress(np.frombuffer(data, dtype=np.uint8), level)[0]):
        rc, out, nb = orc.decompress(stream, len(data) + 16)
        assert rc == orc.D_OK
        assert out == data
        assert bz2.decompress(stream) == data


def test_stage_outputs_chain():
    data = corpus.mixed(1, 5).tobytes()[:400000]
    stream = bz2.compress(data, 2)
    pos, out = 32, b""
    while True:
        rc, bwt_, orig, crc, end = orc.decode_block(stream, pos, 200000)
        if rc == orc.D_FORMAT:
            break
        assert rc == orc.D_OK
        rc, tt = orc.inverse_bwt(bwt_, orig)
        assert rc == orc.D_OK
        rc, o, got = orc.unrle1(tt, 1 << 22)
        assert rc == orc.D_OK and got == crc
        out += o.tobytes()
        pos = end
    assert out == data


def test_errors():
    data = corpus.text(150000, 3).tobytes()
    s = bytearray(bz2.compress(data, 1))
    cap = len(data) + 16
    assert orc.decompress(bytes(s[:3]), cap)[0] == orc.D_INSUFFICIENT
    bad = bytearray(s); bad[0] = ord("C")
    assert orc.decompress(bytes(bad), cap)[0] == orc.D_HEADER
    bad = bytearray(s); bad[3] = ord("0")
    assert orc.decompress(bytes(bad), cap)[0] == orc.D_HEADER
    bad = bytearray(s); bad[5] ^= 1                      # block magic
    assert orc.decompress(bytes(bad), cap)[0] == orc.D_FORMAT
    bad = bytearray(s); bad[11] ^= 0x10                  # stored block CRC
    rc, out, nb = orc.decompress(bytes(bad), cap)
    assert rc == orc.D_BLOCK_CRC and out == data[:len(out)] and len(out) > 0
    bad = bytearray(s); bad[-2] ^= 0x40                  # stream CRC (last bytes)
    rc, out, nb = orc.decompress(bytes(bad), cap)
    assert rc in (orc.D_STREAM_CRC, orc.D_FORMAT) and out == data
    assert orc.decompress(bytes(s[:len(s) // 2]), cap)[0] in (orc.D_INSUFFICIENT, orc.D_DECODE, orc.D_BLOCK_SIZE)
    assert orc.decompress(bytes(s), 1000)[0] == orc.D_CAPACITY
    # trailing bytes after the end marker are ignored (S/BZip2InputStream.swift:228-236)
    rc, out, nb = orc.decompress(bytes(s) + b"garbage", cap)
    assert rc == orc.D_OK and out == data


def test_fuzzed_streams_never_crash():
    rng = np.random.default_rng(11)
    data = corpus.mixed(1, 8).tobytes()[:120000]
    s = bz2.compress(data, 1)
    for _ in range(300):
        bad = bytearray(s)
        for _ in range(int(rng.integers(1, 4))):
            bad[int(rng.integers(4, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        rc, out, nb = orc.decompress(bytes(bad), len(data) * 60)
        if rc == orc.D_OK:
            assert out == data
