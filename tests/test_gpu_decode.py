"""GPU decoder (kd_decode.hip) against the decoder oracle and libbz2, through the
C ABI.  Bit-exact bytes; error codes and the number of bytes handed out before the error follow the
oracle (= the reference's order of checks)."""
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

pytestmark = pytest.mark.gpu

# oracle code -> C ABI code
MAP = {0: 0, -1: -6, -2: -7, -3: -8, -4: -9, -5: -10, -6: -11, -7: -12, -8: -13, -9: -14, -10: -15, -11: -4}


@pytest.fixture(scope="module")
def ctx():
    from bzip2_amd import Context
    c = Context()
    yield c
    c.close()


def _inputs():
    rng = np.random.default_rng(7)
    yield "empty", b""
    yield "one", b"x"
    yield "text", corpus.text(300000, 1).tobytes()
    yield "random", rng.integers(0, 256, 250000, dtype=np.uint8).tobytes()
    yield "runs", (b"a" * 1000 + b"b" * 4 + b"c" * 5 + b"d" * 259 + b"e" * 3) * 300
    yield "zeros", bytes(3 << 20)
    yield "ff255", b"\xff" * 255 * 40
    yield "mixed", corpus.mixed(3, 99).tobytes()
    yield "count_eq_byte", (b"\x05" * 9 + b"\x00" * 4 + b"\x04" * 8) * 1000
    yield "abab", b"ab" * 300000
    yield "fours", bytes(np.repeat(rng.integers(0, 256, 200000, dtype=np.uint8), 4))


@pytest.mark.parametrize("name,data", list(_inputs()), ids=[n for n, _ in _inputs()])
@pytest.mark.parametrize("level", [1, 9])
def test_decode_equals_libbz2(ctx, name, data, level):
    for stream in (bz2.compress(data, level), ctx.compress(data, level)):
        rc, out = ctx.decompress(stream, len(data) + 16)
        assert rc == 0
        assert out == data


def test_large_roundtrip(ctx):
    data = corpus.mixed(32, 1234).tobytes()
    stream = ctx.compress(data, 9)
    rc, out = ctx.decompress(stream, len(data))
    assert rc == 0 and out == data
    rc, out = ctx.decompress(stream, len(data) - 1)
    assert rc == -4 and len(out) > 0 and out == data[:len(out)]


def test_more_blocks_than_one_batch(ctx):
    # > 1024 blocks: the chain walk continues across candidate batches
    data = corpus.mixed(160, 77).tobytes()
    stream = ctx.compress(data, 1)
    rc, out = ctx.decompress(stream, len(data))
    assert rc == 0 and out == data
    assert ctx.last_stats()["blocks"] > 1024
    bad = bytearray(stream)
    bad[len(bad) * 3 // 4] ^= 8   # damage in a later batch: the prefix is delivered
    rc, out = ctx.decompress(bytes(bad), len(data))
    # (a block whose CRC fails has been handed out too: only what precedes it is the input)
    a, b = np.frombuffer(out, dtype=np.uint8), np.frombuffer(data, dtype=np.uint8)[:len(out)]
    diff = np.flatnonzero(a != b[:a.size])
    first = int(diff[0]) if diff.size else a.size
    assert rc != 0 and first > len(data) // 2


def test_errors_follow_the_oracle(ctx):
    data = corpus.mixed(1, 3).tobytes()[:350000]
    s = bz2.compress(data, 1)
    cap = len(data) + 16
    cases = [s[:3], b"CZh9" + s[4:], s[:3] + b"0" + s[4:], s + b"garbage", s[:len(s) // 2], s[:-3], s[:-11]]
    for pos, bit in ((5, 1), (11, 0x10), (len(s) - 2, 0x40), (20, 1), (40, 0x80), (len(s) // 2, 4), (15, 2)):
        bad = bytearray(s)
        bad[pos] ^= bit
        cases.append(bytes(bad))
    for k, st in enumerate(cases):
        orc_rc, orc_out, _ = orc.decompress(st, cap)
        rc, out = ctx.decompress(st, cap)
        assert rc == MAP[orc_rc], (k, rc, orc_rc)
        assert out == orc_out, k


def test_back_to_back_block_magics_get_the_reference_error(ctx):
    """the 48-bit block magic can recur every 6 bytes (every 45 bits, even: it overlaps
    itself by 3), far more often than real blocks.  A crafted stream of nothing but magics must
    fail like the reference (its first block is garbage), not with "too many candidates"."""
    magic = bytes.fromhex("314159265359")
    for n_magic in (10, 40000):
        st = b"BZh9" + magic * n_magic
        orc_rc, orc_out, _ = orc.decompress(st, 1 << 20)
        rc, out = ctx.decompress(st, 1 << 20)
        assert orc_rc != 0 and rc == MAP[orc_rc], (n_magic, rc, orc_rc)
        assert out == orc_out
    # packed at the 45-bit self-overlap: 0x314159265359 shifted so that its last 3 bits start the next
    bits = format(0x314159265359, "048b")
    packed = (bits[:45] * 30000) + bits[45:]
    packed += "0" * (-len(packed) % 8)
    st = b"BZh9" + int(packed, 2).to_bytes(len(packed) // 8, "big")
    orc_rc, orc_out, _ = orc.decompress(st, 1 << 20)
    rc, out = ctx.decompress(st, 1 << 20)
    assert orc_rc != 0 and rc == MAP[orc_rc] and out == orc_out


def test_bit_flip_fuzz(ctx):
    rng = np.random.default_rng(5)
    data = corpus.mixed(1, 8).tobytes()[:250000]
    s = bz2.compress(data, 1)
    cap = len(data) * 2
    for k in range(150):
        bad = bytearray(s)
        for _ in range(int(rng.integers(1, 3))):
            bad[int(rng.integers(4, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        orc_rc, orc_out, _ = orc.decompress(bytes(bad), cap)
        rc, out = ctx.decompress(bytes(bad), cap)
        if orc_rc == orc.D_CAPACITY:
            assert rc == -4
            continue
        assert rc == MAP[orc_rc], (k, rc, orc_rc)
        assert out == orc_out, k


def test_cli_roundtrip(tmp_path):
    """python -m bzip2_amd: file -> file.bz2 (readable by libbz2) -> file"""
    import subprocess
    data = corpus.mixed(2, 4).tobytes()
    src = tmp_path / "sample.bin"
    src.write_bytes(data)
    env = dict(os.environ, PYTHONPATH=ROOT)
    subprocess.check_call([sys.executable, "-m", "bzip2_amd", "-5", "-k", str(src)], env=env, cwd=ROOT)
    z = (tmp_path / "sample.bin.bz2").read_bytes()
    assert z[:4] == b"BZh5" and bz2.decompress(z) == data
    src.unlink()
    subprocess.check_call([sys.executable, "-m", "bzip2_amd", "-d", str(tmp_path / "sample.bin.bz2")], env=env, cwd=ROOT)
    assert src.read_bytes() == data and not (tmp_path / "sample.bin.bz2").exists()
    out = subprocess.run([sys.executable, "-m", "bzip2_amd", "-c"], input=data, capture_output=True, env=env, cwd=ROOT)
    assert out.returncode == 0 and bz2.decompress(out.stdout) == data
    # concatenated streams (cat a.bz2 b.bz2 c.bz2), different levels and encoders, trailing garbage
    parts = [data[:700000], b"", data[700000:], b"tail " * 1000]
    cat = bz2.compress(parts[0], 3) + out.stdout[:0] + bz2.compress(parts[1], 9) + \
        subprocess.run([sys.executable, "-m", "bzip2_amd", "-1", "-c"], input=parts[2], capture_output=True, env=env,
                       cwd=ROOT).stdout + bz2.compress(parts[3], 5) + b"\x00\x00garbage"
    back = subprocess.run([sys.executable, "-m", "bzip2_amd", "-d", "-c"], input=cat, capture_output=True, env=env, cwd=ROOT)
    assert back.returncode == 0 and back.stdout == b"".join(parts)


def test_chain_longer_than_32bit_offsets_is_written_in_passes(orc):
    """Output offsets inside one decoder pass are 32-bit; a chain that expands to more (zeros: 46 MB
    per -9 block) is written in several passes.  The pass size is lowered here (2 MiB) so that a
    30 MB stream of long runs needs many passes; the bytes and the error behaviour stay the same."""
    import os
    import bz2
    import bzip2_amd
    os.environ["BZ2HIP_DEC_PASS_MIB"] = "2"
    try:
        c = bzip2_amd.Context()
    finally:
        del os.environ["BZ2HIP_DEC_PASS_MIB"]
    try:
        rng = np.random.default_rng(8)
        d = np.concatenate((np.zeros(9_000_000, dtype=np.uint8), rng.integers(0, 256, 700_000, dtype=np.uint8),
                            np.repeat(rng.integers(0, 256, 80_000, dtype=np.uint8), 255)))
        for level in (1, 9):
            z = bz2.compress(d.tobytes(), level)
            rc, out = c.decompress(z, d.size)
            assert rc == 0 and out == d.tobytes(), level
            # too small a buffer is still E_CAP with the blocks that fit delivered, as the oracle says
            rc, out = c.decompress(z, 5_000_000)
            orc_rc, orc_out, _ = orc.decompress(z, 5_000_000)
            assert rc == -4 and rc == MAP[orc_rc] and out == orc_out, level
    finally:
        c.close()


# ---- concatenated streams in one call (bz2hip_decompress_streams)
def _streams_model(pieces, cap):
    """What constructing one BZip2InputStream after the other over the joined pieces gives, from the
    decoder oracle run per stream at its known offset: (rc, bytes, streams decoded completely)."""
    z = b"".join(pieces)
    out, off, level = b"", 0, None
    for k, p in enumerate(pieces):
        rest = z[off:]
        if level is None:
            level = rest[3:4]
        elif len(rest) < 4 or rest[:3] != b"BZh" or rest[3:4] != level:
            return 0, out, k, off
        rc, got, _ = orc.decompress(rest, cap - len(out))
        out += got
        if rc != 0:
            return MAP[rc], out, k, off
        off += len(p)
    return 0, out, len(pieces), off


def test_concatenated_streams_decode_in_one_call(ctx):
    rng = np.random.default_rng(77)
    data = corpus.mixed(3, 21).tobytes()
    parts = [data[:1200000], b"", data[1200000:1200007], data[1200007:2500000], b"x" * 70000, data[2500000:]]
    pieces = [ctx.compress(p, 9) if k % 2 else bz2.compress(p, 9) for k, p in enumerate(parts)]
    z = b"".join(pieces)
    rc, out, ns = ctx.decompress_streams(z, len(data) + 70000)
    assert rc == 0 and ns == len(parts) and out == b"".join(parts) == bz2.decompress(z)
    assert ctx.decode_consumed() == len(z)
    # the single-stream call still stops at the first end marker, like the reference
    rc, one = ctx.decompress(z, len(data))
    assert rc == 0 and one == parts[0] and ctx.decode_consumed() == len(pieces[0])
    # trailing bytes that are no stream: ignored, and reported through decode_consumed
    for junk in (b"\x00", b"garbage after the end", b"BZ", b"BZh", b"BZx9" + pieces[1][4:]):
        rc, out, ns = ctx.decompress_streams(z + junk, len(data) + 70000)
        assert rc == 0 and ns == len(parts) and out == b"".join(parts), junk[:8]
        assert ctx.decode_consumed() == len(z)
    # a stream of another level ends the call; the caller goes on from decode_consumed
    other = [bz2.compress(parts[0], 5), ctx.compress(parts[3], 5)]
    zz = pieces[0] + pieces[1] + other[0] + other[1] + pieces[4]
    rc, out, ns = ctx.decompress_streams(zz, len(data))
    used = ctx.decode_consumed()
    assert rc == 0 and ns == 2 and out == parts[0] and used == len(pieces[0]) + len(pieces[1])
    rc, out, ns = ctx.decompress_streams(zz[used:], len(data))
    assert rc == 0 and ns == 2 and out == parts[0] + parts[3]
    used += ctx.decode_consumed()
    rc, out, ns = ctx.decompress_streams(zz[used:], len(data))
    assert rc == 0 and ns == 1 and out == parts[4] and used + ctx.decode_consumed() == len(zz)
    del rng


def test_concatenated_streams_errors_are_those_of_the_failing_stream(ctx):
    rng = np.random.default_rng(78)
    data = corpus.mixed(1, 5).tobytes()
    parts = [data[:260000], data[260000:300000], data[300000:650000], data[650000:700000]]
    pieces = [bz2.compress(p, 1) for p in parts]
    cap = 800000
    offs = np.cumsum([0] + [len(p) for p in pieces])
    cases = []
    for j in range(len(pieces)):
        n = len(pieces[j])
        for pos, bit in ((n - 1, 1), (n - 3, 0x20), (n - 9, 4), (n // 2, 8), (12, 1), (5, 0x10), (3, 1), (0, 2)):
            bad = list(pieces)
            b = bytearray(bad[j]); b[pos] ^= bit; bad[j] = bytes(b)
            cases.append((f"flip s{j} @{pos}", bad))
        for _ in range(6):
            bad = list(pieces)
            b = bytearray(bad[j])
            for _ in range(int(rng.integers(1, 3))):
                b[int(rng.integers(4, n))] ^= 1 << int(rng.integers(0, 8))
            bad[j] = bytes(b)
            cases.append((f"rand s{j}", bad))
    cases.append(("cut in the last trailer", pieces[:-1] + [pieces[-1][:-2]]))
    cases.append(("cut in a block of the last stream", pieces[:-1] + [pieces[-1][:len(pieces[-1]) // 2]]))
    cases.append(("only a header follows", pieces + [b"BZh1"]))
    cases.append(("header and half a magic follow", pieces + [b"BZh1\x31\x41\x59"]))
    for name, bad in cases:
        want_rc, want_out, want_ns, want_used = _streams_model(bad, cap)
        rc, out, ns = ctx.decompress_streams(b"".join(bad), cap)
        if want_rc == MAP[orc.D_CAPACITY]:
            assert rc == -4, name
            continue
        assert (rc, ns) == (want_rc, want_ns), (name, rc, ns, want_rc, want_ns)
        assert out == want_out, name
        assert ctx.decode_consumed() == want_used, name
    # output buffer that ends inside stream 2: E_CAP with the blocks that fit
    for cap2 in (offs[0] + 10, 260000, 260000 + 39999, 300000 + 120000, 699999):
        cap2 = int(cap2)
        want_rc, want_out, want_ns, _ = _streams_model(pieces, cap2)
        rc, out, ns = ctx.decompress_streams(b"".join(pieces), cap2)
        assert want_rc == MAP[orc.D_CAPACITY] and (rc, ns) == (-4, want_ns) and out == want_out, cap2


def test_many_small_streams_share_the_batches(ctx):
    """3000 streams of a few KB (more candidates than one batch holds): one scan, three batches."""
    import torch
    rng = np.random.default_rng(79)
    data = corpus.mixed(8, 31).tobytes()
    cuts = np.sort(rng.choice(len(data), 2999, replace=False))
    parts = [data[a:b] for a, b in zip([0] + list(cuts), list(cuts) + [len(data)])]
    pieces = [bz2.compress(p, 9) for p in parts]
    z = b"".join(pieces)
    rc, out, ns = ctx.decompress_streams(z, len(data))
    assert rc == 0 and ns == 3000 and out == data
    assert ctx.last_stats()["blocks"] >= 3000
    # device form, and damage in a late stream: everything before it is delivered
    d_in = torch.frombuffer(bytearray(z), dtype=torch.uint8).cuda()
    d_out = torch.empty(len(data), dtype=torch.uint8, device="cuda")
    rc, got, ns = ctx.decompress_streams_device(d_in.data_ptr(), d_in.numel(), d_out.data_ptr(), d_out.numel())
    assert rc == 0 and got == len(data) and ns == 3000 and d_out.cpu().numpy().tobytes() == data
    bad = bytearray(z)
    bad[len(z) * 9 // 10] ^= 0x10
    rc, out, ns = ctx.decompress_streams(bytes(bad), len(data))
    ends = np.cumsum([len(p) for p in pieces])
    assert rc != 0 and ns == int(np.searchsorted(ends, len(z) * 9 // 10, side="right")) and ns > 2048
    done = sum(len(p) for p in parts[:ns])
    assert out[:done] == data[:done]
