"""GPU test of the C++ host mirror of the reference classes (include/bzip2javapi.hpp):
BZip2OutputStream / BZip2BlockCompressor / BZip2BitOutputStream used as user code would use the
reference, output compared byte-for-byte with the oracle."""
import os
import subprocess

import numpy as np
import pytest

from bzip2_amd import corpus

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "cpp", "javapi_driver")


def _run(mode, level, data, tmp_path):
    assert os.path.exists(DRIVER), "build tests/cpp (python __graft_entry__.py)"
    fin = tmp_path / "in.bin"
    fout = tmp_path / "out.bz2"
    fin.write_bytes(data)
    subprocess.run([DRIVER, mode, str(level), str(fin), str(fout)], check=True, timeout=120)
    return fout.read_bytes()


def _data():
    return np.concatenate((corpus.text(260000, 8), corpus.segment("runs", 150000, 9),
                           np.zeros(70000, dtype=np.uint8),
                           corpus.segment("random", 50001, 10))).tobytes()


@pytest.mark.parametrize("mode", ["stream", "bytewise", "block"])
def test_javapi_modes(orc, tmp_path, mode):
    d = _data()
    for level in (1, 9):
        want, _ = orc.compress(d, level)
        assert _run(mode, level, d, tmp_path) == want, (mode, level)


def test_javapi_empty(orc, tmp_path):
    for mode in ("stream", "block"):
        assert _run(mode, 9, b"", tmp_path) == bytes.fromhex("425a683917724538509000000000")


def test_javapi_errors():
    out = subprocess.run([DRIVER, "errors"], check=True, timeout=60, capture_output=True).stdout.decode()
    assert "level 0: IllegalArgumentException Invalid BZip2 block size 0" in out
    assert "level 10: IllegalArgumentException Invalid BZip2 block size 10" in out
    assert "header bytes at construction: 4" in out
    assert "write after finish: IOException Write beyond end of stream" in out
    assert "sink closed: 1" in out


def _decode(mode, stream, tmp_path):
    fin = tmp_path / "in.bz2"
    fout = tmp_path / "out.bin"
    fin.write_bytes(stream)
    r = subprocess.run([DRIVER, mode, "0", str(fin), str(fout)], check=True, timeout=120, capture_output=True)
    return r.stdout.decode(), fout.read_bytes()


@pytest.mark.parametrize("mode", ["decode", "decode1"])
def test_javapi_input_stream(orc, tmp_path, mode):
    """BZip2InputStream (S/BZip2InputStream.swift:45-258) through the C++ mirror"""
    import bz2
    d = _data()
    for stream in (bz2.compress(d, 9), orc.compress(d, 1)[0]):
        log, out = _decode(mode, stream, tmp_path)
        assert out == d and f"ok {len(d)}" in log
        assert "after end: -1" in log and "source closed: 1" in log
        assert "read after close: IOException Stream closed" in log


def test_javapi_input_stream_errors(orc, tmp_path):
    import bz2
    d = _data()
    s = bytearray(bz2.compress(d, 1))
    bad = bytearray(s); bad[11] ^= 0x10          # stored CRC of the first block
    log, out = _decode("decode", bytes(bad), tmp_path)
    rc, want, _ = orc.decompress(bytes(bad), len(d) + 16)
    assert rc == orc.D_BLOCK_CRC and out == want
    assert f"IOException BZip2 block CRC error after {len(want)}" in log and "after end: -1" in log
    log, out = _decode("decode", b"BZh0" + bytes(s[4:]), tmp_path)
    assert "IOException Invalid BZip2 header after 0" in log
    log, out = _decode("decode", bytes(s[:len(s) // 2]), tmp_path)
    rc, want, _ = orc.decompress(bytes(s[:len(s) // 2]), len(d) + 16)
    assert out == want and "IOException" in log and f"after {len(want)}" in log
