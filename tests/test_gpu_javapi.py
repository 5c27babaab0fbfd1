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
