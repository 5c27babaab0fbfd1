"""The allocation-free entry point (bz2hip_compress_device_noalloc, include/bz2hip.h): called through
the C ABI, its stream must be byte-identical to bz2hip_compress_device's -- on two calls in a row, the
second with a smaller input (the static scratch keeps the first call's contents) -- and sizes beyond
its maximum are refused.  Collected last (file name), so a failure here hides no other test."""
import bz2
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(ctx, torch, data, level):
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(data).to(dev) if data.size else torch.zeros(4, dtype=torch.uint8, device=dev)
    cap = ctx.L.bz2hip_compress_bound(data.size)
    d_a = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
    d_b = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
    na = ctx.compress_device(d_in.data_ptr(), data.size, level, d_a.data_ptr(), cap)
    nb = ctx.compress_device_noalloc(d_in.data_ptr(), data.size, level, d_b.data_ptr(), cap)
    a = d_a[:na].cpu().numpy().tobytes()
    b = d_b[:nb].cpu().numpy().tobytes()
    return a, b


def test_noalloc_matches_compress_device_two_calls(ctx):
    import torch
    from bzip2_amd import corpus
    big = np.concatenate((corpus.text(400000, 11), corpus.segment("runs", 300000, 12),
                          np.zeros(200000, dtype=np.uint8), corpus.segment("random", 148576, 13)))
    assert big.size == 1 << 20  # the stated maximum
    small = np.concatenate((corpus.segment("markov", 90000, 14), np.tile(np.frombuffer(b"abc", dtype=np.uint8), 5000)))
    for level in (9, 1):
        for data in (big, small):       # second call: smaller input, same scratch
            a, b = _both(ctx, torch, data, level)
            assert b == a, (level, data.size)
            assert bz2.decompress(b) == data.tobytes()
    a, b = _both(ctx, torch, np.zeros(0, dtype=np.uint8), 9)
    assert a == b == bytes.fromhex("425a683917724538509000000000")


def test_noalloc_rejects_sizes_beyond_the_maximum(ctx):
    import torch
    dev = torch.device("cuda", 0)
    n = (1 << 20) + 1
    d_in = torch.zeros(n, dtype=torch.uint8, device=dev)
    d_out = torch.empty(n + 4096, dtype=torch.uint8, device=dev)
    got = C.c_size_t(0)
    rc = ctx.L.bz2hip_compress_device_noalloc(ctx.h, C.c_void_p(d_in.data_ptr()), n, 9,
                                              C.c_void_p(d_out.data_ptr()), n + 4096, C.byref(got))
    assert rc == -1
    assert b"BZ2HIP_NOALLOC_MAX_INPUT" in ctx.L.bz2hip_last_error(ctx.h)
    # the context still works through the allocating entry point afterwards
    data = np.frombuffer(b"hello hello hello", dtype=np.uint8)
    assert bz2.decompress(ctx.compress(data, 9)) == data.tobytes()


def test_noalloc_really_allocates_nothing():
    """The property in the name: free device memory is the same before and after the call.  A FRESH
    context (its own batch workspace has never been allocated, so the call cannot be living off an
    earlier allocating call) makes one warm-up call -- the HIP runtime loads code objects and sizes
    its private-segment scratch on first launches -- and then free memory must not move over two
    more calls, the second with a smaller input."""
    import torch
    from bzip2_amd import Context, corpus
    dev = torch.device("cuda", 0)
    c = Context()
    try:
        big = corpus.mixed(1, 21)
        small = corpus.text(70000, 22)
        d_big = torch.from_numpy(big).to(dev)
        d_small = torch.from_numpy(small).to(dev)
        cap = c.L.bz2hip_compress_bound(big.size)
        d_out = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
        n0 = c.compress_device_noalloc(d_big.data_ptr(), big.size, 9, d_out.data_ptr(), cap)   # warm-up
        first = d_out[:n0].cpu().numpy().tobytes()
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info()
        n1 = c.compress_device_noalloc(d_big.data_ptr(), big.size, 9, d_out.data_ptr(), cap)
        free1, _ = torch.cuda.mem_get_info()
        n2 = c.compress_device_noalloc(d_small.data_ptr(), small.size, 9, d_out.data_ptr(), cap)
        free2, _ = torch.cuda.mem_get_info()
        assert free1 == free0 and free2 == free0, (free0, free1, free2)
        assert n1 == n0
        assert bz2.decompress(first) == big.tobytes()
        assert bz2.decompress(d_out[:n2].cpu().numpy().tobytes()) == small.tobytes()
    finally:
        c.close()
