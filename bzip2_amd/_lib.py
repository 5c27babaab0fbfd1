"""ctypes binding of libbz2hip.so (the C ABI in include/bz2hip.h).

The product path: there is NO CPU fallback.  If the HIP library is missing or no GPU is
present, every compute entry point raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("BZ2HIP_LIB") or os.path.join(_HERE, "csrc", "libbz2hip.so")  # override: experiments only

OK, E_PARAM, E_NOMEM, E_HIP, E_CAP, E_STATE = 0, -1, -2, -3, -4, -5
MAX_ALPHA, MAX_TABLES = 258, 6

# every symbol include/bz2hip.h declares
SYMBOLS = [
    "bz2hip_create", "bz2hip_destroy", "bz2hip_last_error", "bz2hip_version",
    "bz2hip_compress_bound", "bz2hip_compress", "bz2hip_compress_device",
    "bz2hip_compress_blocks_device", "bz2hip_compress_blocks", "bz2hip_stitch",
    "bz2hip_stream_begin", "bz2hip_stream_write", "bz2hip_stream_finish", "bz2hip_stream_read",
    "bz2hip_stream_pending", "bz2hip_rle1_split", "bz2hip_bwt", "bz2hip_mtf_rle2",
    "bz2hip_huffman", "bz2hip_block_compress", "bz2hip_profile_reset", "bz2hip_profile_get", "bz2hip_last_stats3",
    "bz2hip_last_stats", "bz2hip_last_stats2", "bz2hip_decompress", "bz2hip_decompress_device",
    "bz2hip_decompress_streams", "bz2hip_decompress_streams_device",
    "bz2hip_compress_device_noalloc", "bz2hip_shard_scan", "bz2hip_shard_chain",
    "bz2hip_shard_compress", "bz2hip_stitch_device", "bz2hip_host_alloc", "bz2hip_host_free",
    "bz2hip_tie_stats", "bz2hip_tie_stats2", "bz2hip_decode_consumed",
]


class Opts(C.Structure):
    _fields_ = [("device", C.c_int), ("max_batch_blocks", C.c_int), ("tie_rule", C.c_int),
                ("profile", C.c_int), ("n_devices", C.c_int), ("devices", C.c_int * 8)]


class BZ2HipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"bz2hip error {code}: {msg}")
        self.code = code


_lib = None


def load():
    """Loads libbz2hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(f"{SO_PATH} missing: build it with `make -C bzip2_amd/csrc` "
                          "(there is no CPU fallback)")
    # PyTorch-ROCm bundles its own libamdhip64; two HIP runtimes in one process do not see the GPU
    # from the second one.  Loading torch's first makes libbz2hip.so resolve to the same runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(SO_PATH)
    vp, sz, u32p = C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)
    L.bz2hip_create.restype = C.c_int
    L.bz2hip_create.argtypes = [C.POINTER(vp), C.POINTER(Opts)]
    L.bz2hip_destroy.restype = None
    L.bz2hip_destroy.argtypes = [vp]
    L.bz2hip_last_error.restype = C.c_char_p
    L.bz2hip_last_error.argtypes = [vp]
    L.bz2hip_version.restype = C.c_char_p
    L.bz2hip_compress_bound.restype = sz
    L.bz2hip_compress_bound.argtypes = [sz]
    L.bz2hip_compress.restype = C.c_int
    L.bz2hip_compress.argtypes = [vp, vp, sz, C.c_int, vp, sz, C.POINTER(sz)]
    L.bz2hip_compress_device.restype = C.c_int
    L.bz2hip_compress_device.argtypes = [vp, vp, sz, C.c_int, vp, sz, C.POINTER(sz)]
    L.bz2hip_compress_blocks_device.restype = C.c_int
    L.bz2hip_compress_blocks_device.argtypes = [vp, vp, sz, C.c_int, vp, sz, vp, vp, C.c_uint32, u32p]
    L.bz2hip_compress_blocks.restype = C.c_int
    L.bz2hip_compress_blocks.argtypes = [vp, vp, sz, C.c_int, vp, sz, vp, vp, C.c_uint32, u32p]
    L.bz2hip_stitch.restype = C.c_int
    L.bz2hip_stitch.argtypes = [C.c_int, C.c_uint32, vp, vp, vp, vp, vp, sz, C.POINTER(sz)]
    L.bz2hip_stream_begin.restype = C.c_int
    L.bz2hip_stream_begin.argtypes = [vp, C.c_int]
    L.bz2hip_stream_write.restype = C.c_int
    L.bz2hip_stream_write.argtypes = [vp, vp, sz]
    L.bz2hip_stream_finish.restype = C.c_int
    L.bz2hip_stream_finish.argtypes = [vp]
    L.bz2hip_stream_read.restype = C.c_int
    L.bz2hip_stream_read.argtypes = [vp, vp, sz, C.POINTER(sz)]
    L.bz2hip_stream_pending.restype = sz
    L.bz2hip_stream_pending.argtypes = [vp]
    L.bz2hip_rle1_split.restype = C.c_int
    L.bz2hip_rle1_split.argtypes = [vp, vp, sz, C.c_int, C.c_uint32, u32p, vp, vp, vp, vp, vp]
    L.bz2hip_bwt.restype = C.c_int
    L.bz2hip_bwt.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, vp, vp, vp, C.c_uint32]
    L.bz2hip_mtf_rle2.restype = C.c_int
    L.bz2hip_mtf_rle2.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp]
    L.bz2hip_huffman.restype = C.c_int
    L.bz2hip_huffman.argtypes = [vp, vp, vp, vp, vp, C.c_uint32, C.c_uint32, vp, vp, vp,
                                 C.c_uint32, vp]
    L.bz2hip_block_compress.restype = C.c_int
    L.bz2hip_block_compress.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, vp, vp, vp, sz, vp, C.c_uint32]
    L.bz2hip_profile_reset.restype = C.c_int
    L.bz2hip_profile_reset.argtypes = [vp]
    L.bz2hip_profile_get.restype = C.c_int
    L.bz2hip_profile_get.argtypes = [vp, C.POINTER(C.c_int), vp, vp, vp, vp]
    L.bz2hip_last_stats.restype = C.c_int
    L.bz2hip_last_stats.argtypes = [vp, vp, vp, vp, vp]
    L.bz2hip_last_stats2.restype = C.c_int
    L.bz2hip_last_stats2.argtypes = [vp, vp, vp]
    L.bz2hip_last_stats3.restype = C.c_int
    L.bz2hip_last_stats3.argtypes = [vp, vp, vp]
    L.bz2hip_decompress.restype = C.c_int
    L.bz2hip_decompress.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz)]
    L.bz2hip_decompress_device.restype = C.c_int
    L.bz2hip_decompress_device.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz)]
    L.bz2hip_decompress_streams.restype = C.c_int
    L.bz2hip_decompress_streams.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz), C.POINTER(C.c_uint32)]
    L.bz2hip_decompress_streams_device.restype = C.c_int
    L.bz2hip_decompress_streams_device.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz), C.POINTER(C.c_uint32)]
    L.bz2hip_compress_device_noalloc.restype = C.c_int
    L.bz2hip_compress_device_noalloc.argtypes = [vp, vp, sz, C.c_int, vp, sz, C.POINTER(sz)]
    L.bz2hip_shard_scan.restype = C.c_int
    L.bz2hip_shard_scan.argtypes = [vp, vp, sz, C.c_int]
    L.bz2hip_shard_chain.restype = C.c_int
    L.bz2hip_shard_chain.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_int, u32p, C.POINTER(C.c_uint64)]
    L.bz2hip_shard_compress.restype = C.c_int
    L.bz2hip_shard_compress.argtypes = [vp, vp, sz, C.POINTER(C.c_uint64), vp, C.c_uint32, u32p]
    L.bz2hip_decode_consumed.restype = C.c_int
    L.bz2hip_decode_consumed.argtypes = [vp, C.POINTER(sz)]
    L.bz2hip_tie_stats.restype = C.c_int
    L.bz2hip_tie_stats.argtypes = [vp, vp, vp, vp]
    L.bz2hip_tie_stats2.restype = C.c_int
    L.bz2hip_tie_stats2.argtypes = [vp, vp]
    L.bz2hip_host_alloc.restype = C.c_int
    L.bz2hip_host_alloc.argtypes = [C.POINTER(vp), sz]
    L.bz2hip_host_free.restype = None
    L.bz2hip_host_free.argtypes = [vp]
    L.bz2hip_stitch_device.restype = C.c_int
    L.bz2hip_stitch_device.argtypes = [vp, C.c_int, C.c_uint32, vp, vp, vp, vp, vp, sz, C.POINTER(sz)]
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class PinnedBuffer:
    """Page-locked host buffer (bz2hip_host_alloc) viewed as a numpy uint8 array (.a)."""

    def __init__(self, nbytes):
        self.L = load()
        self.p = C.c_void_p()
        rc = self.L.bz2hip_host_alloc(C.byref(self.p), nbytes)
        if rc != OK:
            raise BZ2HipError(rc, "bz2hip_host_alloc failed")
        self.a = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(self.p.value))

    def free(self):
        if self.p:
            self.a = None
            self.L.bz2hip_host_free(self.p)
            self.p = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def stitch(level, segments):
    """segments: list of (bits: uint8 array/bytes, nbits, crcs: uint32 array) in block order.
    Returns the complete .bz2 stream (host-side ordered concatenation, bz2hip_stitch)."""
    L = load()
    n = len(segments)
    bits = [np.ascontiguousarray(np.frombuffer(bytes(s[0]), dtype=np.uint8) if not isinstance(s[0], np.ndarray) else s[0], dtype=np.uint8) for s in segments]
    bits = [b if b.size else np.zeros(1, dtype=np.uint8) for b in bits]
    crcs = [np.ascontiguousarray(s[2], dtype=np.uint32) for s in segments]
    crcs_nz = [c if c.size else np.zeros(1, dtype=np.uint32) for c in crcs]
    pb = (C.c_void_p * max(n, 1))(*[b.ctypes.data for b in bits])
    pc = (C.c_void_p * max(n, 1))(*[c.ctypes.data for c in crcs_nz])
    nbits = np.array([int(s[1]) for s in segments] + [0], dtype=np.uint64)
    ncrc = np.array([c.size for c in crcs] + [0], dtype=np.uint32)
    cap = int(nbits.sum() // 8) + 64
    out = np.empty(cap, dtype=np.uint8)
    got = C.c_size_t(0)
    rc = L.bz2hip_stitch(level, n, pb, _p(nbits), pc, _p(ncrc), _p(out), cap, C.byref(got))
    if rc != OK:
        raise BZ2HipError(rc, "bz2hip_stitch failed")
    return out[: got.value].tobytes()


class Context:
    """One GPU compression context (bz2hip_ctx)."""

    def __init__(self, device=-1, max_batch_blocks=0, tie_rule=0, profile=False, devices=None):
        """devices: list of HIP ordinals -> the host-buffer entry points shard every stream over them"""
        self.L = load()
        self.h = C.c_void_p()
        o = Opts(device, max_batch_blocks, tie_rule, int(profile))
        if devices:
            o.n_devices = len(devices)
            for i, d in enumerate(devices):
                o.devices[i] = d
        rc = self.L.bz2hip_create(C.byref(self.h), C.byref(o))
        if rc != OK:
            raise BZ2HipError(rc, "bz2hip_create failed: " +
                              self.L.bz2hip_last_error(None).decode())

    def close(self):
        if self.h:
            self.L.bz2hip_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != OK:
            raise BZ2HipError(rc, self.L.bz2hip_last_error(self.h).decode())

    # ---- whole stream
    def compress(self, data, level=9):
        a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else \
            np.ascontiguousarray(data, dtype=np.uint8)
        cap = self.L.bz2hip_compress_bound(a.size)
        out = np.empty(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        self._chk(self.L.bz2hip_compress(self.h, _p(a), a.size, level, _p(out), cap, C.byref(n)))
        return out[: n.value].tobytes()

    def compress_into(self, a, level, out):
        """host arrays in / out (numpy uint8, e.g. PinnedBuffer.a) -> compressed length; no copies"""
        n = C.c_size_t(0)
        self._chk(self.L.bz2hip_compress(self.h, _p(a), a.size, level, _p(out), out.size, C.byref(n)))
        return n.value

    # ---- decoder: (rc, bytes the reference would have handed out before failing); rc 0 = whole stream
    def decompress(self, stream, cap):
        a = np.frombuffer(bytes(stream), dtype=np.uint8)
        out = np.empty(max(cap, 1), dtype=np.uint8)
        n = C.c_size_t(0)
        rc = self.L.bz2hip_decompress(self.h, _p(a), a.size, _p(out), cap, C.byref(n))
        if rc in (-1, -2, -3):
            self._chk(rc)
        return rc, out[: n.value].tobytes()

    def decompress_streams(self, data, cap):
        """concatenated streams of one level -> (rc, bytes, streams decoded completely)"""
        a = np.frombuffer(bytes(data), dtype=np.uint8)
        out = np.empty(max(cap, 1), dtype=np.uint8)
        n = C.c_size_t(0)
        ns = C.c_uint32(0)
        rc = self.L.bz2hip_decompress_streams(self.h, _p(a), a.size, _p(out), cap, C.byref(n), C.byref(ns))
        if rc in (-1, -2, -3):
            self._chk(rc)
        return rc, out[: n.value].tobytes(), ns.value

    def decompress_streams_device(self, d_in_ptr, n, d_out_ptr, cap):
        got = C.c_size_t(0)
        ns = C.c_uint32(0)
        rc = self.L.bz2hip_decompress_streams_device(self.h, C.c_void_p(d_in_ptr), n, C.c_void_p(d_out_ptr),
                                                     cap, C.byref(got), C.byref(ns))
        if rc in (-1, -2, -3):
            self._chk(rc)
        return rc, got.value, ns.value

    def decode_consumed(self):
        """input bytes the last successfully decoded stream occupied (concatenated streams continue there)"""
        n = C.c_size_t(0)
        self.L.bz2hip_decode_consumed(self.h, C.byref(n))
        return n.value

    def decompress_device(self, d_in_ptr, n, d_out_ptr, cap):
        got = C.c_size_t(0)
        rc = self.L.bz2hip_decompress_device(self.h, C.c_void_p(d_in_ptr), n, C.c_void_p(d_out_ptr), cap,
                                             C.byref(got))
        if rc in (-1, -2, -3):
            self._chk(rc)
        return rc, got.value

    def compress_device(self, d_in_ptr, n, level, d_out_ptr, cap):
        got = C.c_size_t(0)
        self._chk(self.L.bz2hip_compress_device(self.h, C.c_void_p(d_in_ptr), n, level,
                                                C.c_void_p(d_out_ptr), cap, C.byref(got)))
        return got.value

    def compress_blocks_device(self, d_in_ptr, n, level, d_out_ptr, cap):
        """-> (nbits, crcs) : header/trailer-less bit segment of whole blocks"""
        mx = n // (level * 50000) + 8
        crcs = np.zeros(mx, dtype=np.uint32)
        nb = C.c_uint32(0)
        nbits = C.c_uint64(0)
        self._chk(self.L.bz2hip_compress_blocks_device(self.h, C.c_void_p(d_in_ptr), n, level,
                                                       C.c_void_p(d_out_ptr), cap, C.byref(nbits),
                                                       _p(crcs), mx, C.byref(nb)))
        return nbits.value, crcs[: nb.value].copy()

    def compress_device_noalloc(self, d_in_ptr, n, level, d_out_ptr, cap):
        """bz2hip_compress_device without any device allocation (static scratch; n <= 1 MiB)"""
        got = C.c_size_t(0)
        self._chk(self.L.bz2hip_compress_device_noalloc(self.h, C.c_void_p(d_in_ptr), n, level,
                                                        C.c_void_p(d_out_ptr), cap, C.byref(got)))
        return got.value

    # ---- one stream sharded over GPUs / processes (device pointers)
    def shard_scan(self, d_in_ptr, avail, level):
        self._chk(self.L.bz2hip_shard_scan(self.h, C.c_void_p(d_in_ptr), avail, level))

    def shard_chain(self, first, limit, final):
        """-> (n_blocks, next): blocks starting in [first, limit); next = where the next shard continues"""
        nb = C.c_uint32(0)
        nxt = C.c_uint64(0)
        self._chk(self.L.bz2hip_shard_chain(self.h, first, limit, int(bool(final)), C.byref(nb), C.byref(nxt)))
        return nb.value, nxt.value

    def shard_compress(self, d_out_ptr, cap, max_blocks):
        """-> (nbits, crcs) of the blocks found by the last shard_chain, bit-abutted from bit 0"""
        crcs = np.zeros(max(max_blocks, 1), dtype=np.uint32)
        nb = C.c_uint32(0)
        nbits = C.c_uint64(0)
        self._chk(self.L.bz2hip_shard_compress(self.h, C.c_void_p(d_out_ptr), cap, C.byref(nbits), _p(crcs),
                                               max_blocks, C.byref(nb)))
        return nbits.value, crcs[: nb.value].copy()

    def stitch_device(self, level, segments, d_out_ptr, cap):
        """segments: list of (device pointer, nbits, crcs uint32 array) in block order -> stream length"""
        n = len(segments)
        crcs = [np.ascontiguousarray(s[2], dtype=np.uint32) for s in segments]
        crcs_nz = [c if c.size else np.zeros(1, dtype=np.uint32) for c in crcs]
        pb = (C.c_void_p * max(n, 1))(*[int(s[0]) for s in segments])
        pc = (C.c_void_p * max(n, 1))(*[c.ctypes.data for c in crcs_nz])
        nbits = np.array([int(s[1]) for s in segments] + [0], dtype=np.uint64)
        ncrc = np.array([c.size for c in crcs] + [0], dtype=np.uint32)
        got = C.c_size_t(0)
        self._chk(self.L.bz2hip_stitch_device(self.h, level, n, pb, _p(nbits), pc, _p(ncrc),
                                              C.c_void_p(d_out_ptr), cap, C.byref(got)))
        return got.value

    def compress_blocks(self, data, level):
        """host data -> (bits bytes, nbits, crcs)"""
        a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else \
            np.ascontiguousarray(data, dtype=np.uint8)
        cap = self.L.bz2hip_compress_bound(a.size)
        out = np.empty(cap, dtype=np.uint8)
        mx = a.size // (level * 50000) + 8
        crcs = np.zeros(mx, dtype=np.uint32)
        nb = C.c_uint32(0)
        nbits = C.c_uint64(0)
        self._chk(self.L.bz2hip_compress_blocks(self.h, _p(a), a.size, level, _p(out), cap,
                                                C.byref(nbits), _p(crcs), mx, C.byref(nb)))
        return out[: (nbits.value + 7) // 8].copy(), nbits.value, crcs[: nb.value].copy()

    def split(self, data, level):
        """block start offsets of the stream (K1 split only)"""
        a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else \
            np.ascontiguousarray(data, dtype=np.uint8)
        mx = a.size // (level * 50000) + 8
        nb = C.c_uint32(0)
        starts = np.zeros(mx, dtype=np.uint64)
        self._chk(self.L.bz2hip_rle1_split(self.h, _p(a), a.size, level, mx, C.byref(nb),
                                           _p(starts), None, None, None, None))
        return starts[: nb.value].copy()

    # ---- streaming
    def stream_begin(self, level):
        self._chk(self.L.bz2hip_stream_begin(self.h, level))

    def stream_write(self, data):
        a = np.ascontiguousarray(data, dtype=np.uint8) if isinstance(data, np.ndarray) else \
            np.frombuffer(data, dtype=np.uint8)
        self._chk(self.L.bz2hip_stream_write(self.h, _p(a), a.size))

    def stream_finish(self):
        self._chk(self.L.bz2hip_stream_finish(self.h))

    def stream_read(self, cap=1 << 20):
        # one landing buffer per context, reused: a fresh array per call costs its page faults again
        buf = getattr(self, "_rbuf", None)
        if buf is None or buf.size < cap:
            buf = self._rbuf = np.empty(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        self._chk(self.L.bz2hip_stream_read(self.h, _p(buf), cap, C.byref(n)))
        return buf[: n.value].tobytes()

    def stream_read_into(self, out, offset=0):
        """reads straight into the caller's uint8 array at `offset` (what a C caller does); -> bytes read"""
        n = C.c_size_t(0)
        self._chk(self.L.bz2hip_stream_read(self.h, C.c_void_p(out.ctypes.data + offset), out.size - offset, C.byref(n)))
        return n.value

    def stream_pending(self):
        return self.L.bz2hip_stream_pending(self.h)

    # ---- stages (batched, host arrays)
    def rle1_split(self, data, level):
        a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else \
            np.ascontiguousarray(data, dtype=np.uint8)
        bs = level * 100000
        mx = a.size // (level * 50000) + 8
        nb = C.c_uint32(0)
        starts = np.zeros(mx, dtype=np.uint64)
        rle = np.zeros((mx, bs + 8), dtype=np.uint8)
        rlen = np.zeros(mx, dtype=np.uint32)
        present = np.zeros((mx, 256), dtype=np.uint8)
        crc = np.zeros(mx, dtype=np.uint32)
        self._chk(self.L.bz2hip_rle1_split(self.h, _p(a), a.size, level, mx, C.byref(nb),
                                           _p(starts), _p(rle), _p(rlen), _p(present), _p(crc)))
        k = nb.value
        return starts[:k], [rle[b, : rlen[b]].copy() for b in range(k)], present[:k], crc[:k]

    def bwt(self, blocks, declared_block_size=0):
        """blocks: list of uint8 arrays -> list of (bwt, orig, rounds)"""
        count = len(blocks)
        stride = max(max(len(b) for b in blocks), 1)
        T = np.zeros((count, stride), dtype=np.uint8)
        n = np.zeros(count, dtype=np.uint32)
        for i, b in enumerate(blocks):
            T[i, : len(b)] = np.frombuffer(bytes(b), dtype=np.uint8) if not isinstance(b, np.ndarray) else b
            n[i] = len(b)
        out = np.zeros((count, stride), dtype=np.uint8)
        orig = np.zeros(count, dtype=np.uint32)
        rounds = np.zeros(count, dtype=np.uint32)
        self._chk(self.L.bz2hip_bwt(self.h, _p(T), _p(n), count, stride, _p(out), _p(orig), _p(rounds),
                                    declared_block_size))
        return [(out[i, : n[i]].copy(), int(orig[i]), int(rounds[i])) for i in range(count)]

    def mtf_rle2(self, bwts, presents):
        count = len(bwts)
        stride = max(max(len(b) for b in bwts), 1)
        T = np.zeros((count, stride), dtype=np.uint8)
        n = np.zeros(count, dtype=np.uint32)
        for i, b in enumerate(bwts):
            T[i, : len(b)] = b
            n[i] = len(b)
        pres = np.ascontiguousarray(np.stack(presents), dtype=np.uint8)
        mtf = np.zeros((count, stride + 2), dtype=np.uint16)
        ml = np.zeros(count, dtype=np.uint32)
        al = np.zeros(count, dtype=np.uint32)
        freq = np.zeros((count, 258), dtype=np.int32)
        self._chk(self.L.bz2hip_mtf_rle2(self.h, _p(T), _p(n), count, stride, _p(pres), _p(mtf),
                                         _p(ml), _p(al), _p(freq)))
        return [(mtf[i, : ml[i]].copy(), int(al[i]), freq[i].copy()) for i in range(count)]

    def huffman(self, mtfs, alphas, freqs):
        count = len(mtfs)
        stride = max(len(m) for m in mtfs)
        M = np.zeros((count, stride), dtype=np.uint16)
        ml = np.zeros(count, dtype=np.uint32)
        for i, m in enumerate(mtfs):
            M[i, : len(m)] = m
            ml[i] = len(m)
        al = np.asarray(alphas, dtype=np.uint32)
        fr = np.ascontiguousarray(np.stack(freqs), dtype=np.int32)
        nt = np.zeros(count, dtype=np.uint32)
        lengths = np.zeros((count, 6, 258), dtype=np.int32)
        ss = 18048
        sel = np.zeros((count, ss), dtype=np.uint8)
        ns = np.zeros(count, dtype=np.uint32)
        self._chk(self.L.bz2hip_huffman(self.h, _p(M), _p(ml), _p(al), _p(fr), count, stride,
                                        _p(nt), _p(lengths), _p(sel), ss, _p(ns)))
        return [(int(nt[i]), lengths[i].copy(), sel[i, : ns[i]].copy()) for i in range(count)]

    def block_compress(self, blocks, presents, crcs, declared_block_size=0):
        count = len(blocks)
        stride = max(len(b) for b in blocks)
        T = np.zeros((count, stride), dtype=np.uint8)
        n = np.zeros(count, dtype=np.uint32)
        for i, b in enumerate(blocks):
            T[i, : len(b)] = b
            n[i] = len(b)
        pres = np.ascontiguousarray(np.stack(presents), dtype=np.uint8)
        crc = np.asarray(crcs, dtype=np.uint32)
        bstride = (stride + stride // 2 + 40000 + 3) & ~3
        bits = np.zeros((count, bstride), dtype=np.uint8)
        nbits = np.zeros(count, dtype=np.uint64)
        self._chk(self.L.bz2hip_block_compress(self.h, _p(T), _p(n), count, stride, _p(pres),
                                               _p(crc), _p(bits), bstride, _p(nbits), declared_block_size))
        return [(bits[i, : (int(nbits[i]) + 7) // 8].copy(), int(nbits[i])) for i in range(count)]

    # ---- profiling
    def profile_reset(self):
        self.L.bz2hip_profile_reset(self.h)

    def profile_get(self):
        cnt = C.c_int(0)
        names = (C.c_char_p * 48)()
        ms = (C.c_double * 48)()
        launches = (C.c_uint64 * 48)()
        ab = (C.c_double * 48)()
        self.L.bz2hip_profile_get(self.h, C.byref(cnt), names, ms, launches, ab)
        return {names[i].decode(): {"ms": ms[i], "launches": launches[i], "alg_bytes": ab[i]}
                for i in range(cnt.value)}

    def tie_stats(self):
        """periodic blocks since the context was created: replays run, results reused, cached keys"""
        v = [C.c_uint64(0) for _ in range(3)]
        self.L.bz2hip_tie_stats(self.h, *[C.byref(x) for x in v])
        rej = C.c_uint64(0)
        self.L.bz2hip_tie_stats2(self.h, C.byref(rej))
        return {"replays": v[0].value, "reused": v[1].value, "cached_keys": v[2].value, "rejected": rej.value}

    def last_stats(self):
        v = [C.c_uint64(0) for _ in range(4)]
        self.L.bz2hip_last_stats(self.h, *[C.byref(x) for x in v])
        w = [C.c_uint64(0), C.c_uint64(0)]
        self.L.bz2hip_last_stats2(self.h, C.byref(w[0]), C.byref(w[1]))
        secs = C.c_double(0.0)
        traps = C.c_uint64(0)
        self.L.bz2hip_last_stats3(self.h, C.byref(secs), C.byref(traps))
        return {"blocks": v[0].value, "rounds_sum": v[1].value, "rle_bytes": v[2].value,
                "periodic_blocks": v[3].value, "n_times_rounds": w[0].value,
                "sorted_elems": w[1].value, "bwt_seconds": secs.value, "tie_traps": traps.value}
