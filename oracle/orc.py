"""ctypes loader for the CPU oracle (oracle/libbz2oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (bzip2_amd) must never import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbz2oracle.so")

MAX_ALPHA = 258
MAX_TABLES = 6


def build(force=False):
    if force or not os.path.exists(_SO) or (
        os.path.exists(os.path.join(_HERE, "bz2_oracle.c"))
        and os.path.getmtime(_SO) < max(os.path.getmtime(os.path.join(_HERE, f))
                                        for f in ("bz2_oracle.c", "divsufsort_restate.c", "bz2_decode_oracle.c", "bz2_oracle.h"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        u8p = C.POINTER(C.c_uint8)
        L.orc_crc32_update.restype = C.c_uint32
        L.orc_crc32_update.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
        L.orc_rle1_block.restype = C.c_size_t
        L.orc_rle1_block.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p,
                                     C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_uint32)]
        L.orc_bwt.restype = None
        L.orc_bwt.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int),
                              C.POINTER(C.c_int)]
        L.orc_bwt_divsufsort.restype = None
        L.orc_bwt_divsufsort.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
        L.orc_divsufsort_bwt.restype = C.c_int
        L.orc_divsufsort_bwt.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.orc_declared_block_size.restype = C.c_int
        L.orc_declared_block_size.argtypes = [C.c_int, C.c_int]
        L.orc_bwt_bruteforce.restype = None
        L.orc_bwt_bruteforce.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int),
                                         C.POINTER(C.c_int)]
        L.orc_mtf_rle2.restype = None
        L.orc_mtf_rle2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int),
                                   C.POINTER(C.c_int), C.c_void_p]
        L.orc_huffman_allocate.restype = None
        L.orc_huffman_allocate.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_huffman_code_lengths.restype = None
        L.orc_huffman_code_lengths.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_huffman_optimise.restype = C.c_int
        L.orc_huffman_optimise.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
        L.orc_huffman_codes.restype = None
        L.orc_huffman_codes.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_block_compress.restype = C.c_uint64
        L.orc_block_compress.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_int,
                                         C.c_void_p]
        L.orc_block_bound.restype = C.c_size_t
        L.orc_block_bound.argtypes = [C.c_int]
        L.orc_compress.restype = C.c_size_t
        L.orc_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                   C.POINTER(C.c_int)]
        L.orc_compress_mt.restype = C.c_size_t
        L.orc_compress_mt.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                      C.c_int, C.POINTER(C.c_int)]
        L.orc_compress_bound.restype = C.c_size_t
        L.orc_compress_bound.argtypes = [C.c_size_t]
        L.orc_split.restype = C.c_int
        L.orc_split.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_int]
        L.orc_divsufsort_traps.restype = C.c_int
        L.orc_divsufsort_traps.argtypes = []
        L.orc_decompress.restype = C.c_int
        L.orc_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                     C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        L.orc_decode_block.restype = C.c_int
        L.orc_decode_block.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_int, C.c_void_p,
                                       C.POINTER(C.c_int), C.POINTER(C.c_uint32),
                                       C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        L.orc_inverse_bwt.restype = C.c_int
        L.orc_inverse_bwt.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]
        L.orc_unrle1.restype = C.c_int
        L.orc_unrle1.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t,
                                 C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]
        _lib = L
    return _lib


def _u8(a):
    a = np.ascontiguousarray(np.frombuffer(a, dtype=np.uint8) if isinstance(a, (bytes, bytearray)) else a,
                             dtype=np.uint8)
    return a


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def crc32(data, crc=0xFFFFFFFF):
    a = _u8(data)
    return lib().orc_crc32_update(crc, _ptr(a), a.size)


def block_crc(data):
    return (~crc32(data)) & 0xFFFFFFFF


def rle1_block(data, block_size):
    """-> (consumed, block bytes (np.uint8[n]), present (np.uint8[256]), crc)"""
    a = _u8(data)
    block = np.zeros(block_size + 8, dtype=np.uint8)
    present = np.zeros(256, dtype=np.uint8)
    blen = C.c_int(0)
    crc = C.c_uint32(0)
    consumed = lib().orc_rle1_block(_ptr(a), a.size, block_size, _ptr(block), C.byref(blen),
                                    _ptr(present), C.byref(crc))
    return consumed, block[: blen.value].copy(), present, crc.value


def bwt(block, declared_block_size=0, tie_rule=0):
    """-> (bwt bytes, orig, period count k).  declared_block_size = level * 100000 of the stream the
    block belongs to (0 = the smallest multiple of 100000 that holds it); tie_rule 0 = the
    reference's sorter (tier 2), 1 / 2 = first / last position of the class (tier 1)."""
    a = _u8(block)
    out = np.zeros(max(a.size, 1), dtype=np.uint8)
    orig = C.c_int(0)
    per = C.c_int(0)
    lib().orc_bwt(_ptr(a), a.size, declared_block_size, tie_rule, _ptr(out), C.byref(orig), C.byref(per))
    return out[: a.size], orig.value, per.value


def bwt_divsufsort(block, declared_block_size=0):
    """Tier 2 (restated reference sorter) -> (bwt bytes, orig)"""
    a = _u8(block)
    t = np.zeros(a.size + 1, dtype=np.uint8)
    t[: a.size] = a
    out = np.zeros(max(a.size, 1), dtype=np.uint8)
    orig = C.c_int(0)
    lib().orc_bwt_divsufsort(_ptr(t), a.size, declared_block_size, _ptr(out), C.byref(orig))
    return out[: a.size], orig.value


def divsufsort_raw(block, sa_len):
    """Tier 2 on an int array of exactly sa_len entries -> (orig, trap).  trap = the reference
    would have indexed past its array (S/BZip2DivSufSort.swift:2589-2592) on this block."""
    a = _u8(block)
    t = np.zeros(a.size + 1, dtype=np.uint8)
    t[: a.size] = a
    if a.size:
        t[a.size] = a[0]
    sa = np.zeros(max(sa_len, a.size + 1), dtype=np.int32)
    trap = C.c_int(0)
    o = lib().orc_divsufsort_bwt(_ptr(t), _ptr(sa), a.size, sa_len, C.byref(trap))
    return o, trap.value


def bwt_bruteforce(block):
    a = _u8(block)
    out = np.zeros(max(a.size, 1), dtype=np.uint8)
    f = C.c_int(0)
    l = C.c_int(0)
    lib().orc_bwt_bruteforce(_ptr(a), a.size, _ptr(out), C.byref(f), C.byref(l))
    return out[: a.size], f.value, l.value


def mtf_rle2(bwt_bytes, present):
    a = _u8(bwt_bytes)
    p = _u8(present)
    mtf = np.zeros(a.size + 2, dtype=np.uint16)
    freq = np.zeros(MAX_ALPHA, dtype=np.int32)
    ml = C.c_int(0)
    al = C.c_int(0)
    lib().orc_mtf_rle2(_ptr(a), a.size, _ptr(p), _ptr(mtf), C.byref(ml), C.byref(al), _ptr(freq))
    return mtf[: ml.value].copy(), al.value, freq


def huffman_allocate(sorted_freq, max_len=20):
    a = np.ascontiguousarray(sorted_freq, dtype=np.int32).copy()
    lib().orc_huffman_allocate(_ptr(a), a.size, max_len)
    return a


def huffman_code_lengths(freq, alpha):
    f = np.ascontiguousarray(freq, dtype=np.int32)
    out = np.zeros(MAX_ALPHA, dtype=np.int32)
    lib().orc_huffman_code_lengths(alpha, _ptr(f), _ptr(out))
    return out[:alpha]


def huffman_optimise(mtf, alpha, freq, want_iters=False):
    """-> (n_tables, lengths[6][258], selectors[, iter_lengths[4][6][258]])"""
    m = np.ascontiguousarray(mtf, dtype=np.uint16)
    f = np.ascontiguousarray(freq, dtype=np.int32)
    lengths = np.zeros((MAX_TABLES, MAX_ALPHA), dtype=np.int32)
    sel = np.zeros((m.size + 49) // 50 + 8, dtype=np.uint8)
    nsel = C.c_int(0)
    it = np.zeros((4, MAX_TABLES, MAX_ALPHA), dtype=np.int32) if want_iters else None
    nt = lib().orc_huffman_optimise(_ptr(m), m.size, alpha, _ptr(f), _ptr(lengths), _ptr(sel),
                                    C.byref(nsel), _ptr(it) if want_iters else None)
    if want_iters:
        return nt, lengths, sel[: nsel.value].copy(), it
    return nt, lengths, sel[: nsel.value].copy()


def huffman_codes(n_tables, alpha, lengths):
    l = np.ascontiguousarray(lengths, dtype=np.int32)
    merged = np.zeros((MAX_TABLES, MAX_ALPHA), dtype=np.int32)
    lib().orc_huffman_codes(n_tables, alpha, _ptr(l), _ptr(merged))
    return merged


def block_compress(rle_block, present, crc, declared_block_size=0, tie_rule=0):
    """-> (bytes holding the bits, nbits)"""
    a = _u8(rle_block)
    p = _u8(present)
    out = np.zeros(lib().orc_block_bound(a.size), dtype=np.uint8)
    nbits = lib().orc_block_compress(_ptr(a), a.size, _ptr(p), crc, declared_block_size, tie_rule, _ptr(out))
    return out[: (nbits + 7) // 8].copy(), nbits


def compress(data, level=9, threads=1, tie_rule=0):
    """Whole stream; every block is sorted in an array of level * 100000 + 1 ints whatever `threads`."""
    a = _u8(data)
    cap = lib().orc_compress_bound(a.size)
    out = np.zeros(cap, dtype=np.uint8)
    nb = C.c_int(0)
    if threads <= 1:
        n = lib().orc_compress(_ptr(a), a.size, level, tie_rule, _ptr(out), cap, C.byref(nb))
    else:
        n = lib().orc_compress_mt(_ptr(a), a.size, level, tie_rule, _ptr(out), cap, threads, C.byref(nb))
    if n == 0:
        raise ValueError("oracle compress failed (bad level or capacity)")
    return out[:n].tobytes(), nb.value


_lib_d23 = None


def compress_d23_swift_text(data, level=9):
    """The whole-stream oracle built with -DORC_D23_SWIFT_TEXT=1 (the Swift text's reading of the seed
    back-off, D23).  Only tests/test_oracle.py and tests/golden/make_golden.py use it."""
    global _lib_d23
    if _lib_d23 is None:
        so = os.path.join(_HERE, "libbz2oracle_d23swift.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _lib_d23 = C.CDLL(so)
        _lib_d23.orc_compress.restype = C.c_size_t
        _lib_d23.orc_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                          C.POINTER(C.c_int)]
    a = _u8(data)
    cap = lib().orc_compress_bound(a.size)
    out = np.zeros(cap, dtype=np.uint8)
    nb = C.c_int(0)
    n = _lib_d23.orc_compress(_ptr(a), a.size, level, 0, _ptr(out), cap, C.byref(nb))
    return out[:n].tobytes(), nb.value


def split(data, level=9):
    a = _u8(data)
    mx = a.size // (level * 80000) + 16
    starts = np.zeros(mx, dtype=np.uint64)
    nb = lib().orc_split(_ptr(a), a.size, level, _ptr(starts), mx)
    return starts[:nb].copy()


def divsufsort_traps():
    """Blocks so far on which the reference's sorter would have indexed past its array."""
    return lib().orc_divsufsort_traps()


# ---- decoder restatement (bz2_decode_oracle.c)
D_OK, D_INSUFFICIENT, D_HEADER, D_FORMAT, D_BLOCK_CRC, D_STREAM_CRC = 0, -1, -2, -3, -4, -5
D_TABLES, D_BLOCK_SIZE, D_START_PTR, D_DECODE, D_RANDOMISED, D_CAPACITY = -6, -7, -8, -9, -10, -11


def decompress(stream, cap):
    """-> (rc, bytes the reference would have handed out before failing, blocks)"""
    a = _u8(stream)
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    produced = C.c_size_t(0)
    nb = C.c_int(0)
    rc = lib().orc_decompress(_ptr(a), a.size, _ptr(out), cap, C.byref(produced), C.byref(nb))
    return rc, out[:produced.value].tobytes(), nb.value


def decode_block(stream, bitpos, block_size):
    """block whose magic starts at bit `bitpos` -> (rc, bwt bytes, orig, crc, end bit)"""
    a = _u8(stream)
    bwt_ = np.zeros(block_size, dtype=np.uint8)
    n = C.c_int(0)
    orig, crc, end = C.c_uint32(0), C.c_uint32(0), C.c_uint64(0)
    rc = lib().orc_decode_block(_ptr(a), a.size, bitpos, block_size, _ptr(bwt_), C.byref(n),
                                C.byref(orig), C.byref(crc), C.byref(end))
    return rc, bwt_[:n.value].copy(), orig.value, crc.value, end.value


def inverse_bwt(bwt_bytes, orig):
    a = _u8(bwt_bytes)
    tt = np.zeros(max(a.size, 1), dtype=np.uint8)
    rc = lib().orc_inverse_bwt(_ptr(a), a.size, orig, _ptr(tt))
    return rc, tt[:a.size]


def unrle1(tt, cap):
    a = _u8(tt)
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    ol, crc = C.c_size_t(0), C.c_uint32(0)
    rc = lib().orc_unrle1(_ptr(a), a.size, _ptr(out), cap, C.byref(ol), C.byref(crc))
    return rc, out[:ol.value], crc.value
