"""Deterministic synthetic corpora for tests and bench.

All generators are seeded numpy (PCG64) and vectorised so that 256 MiB builds in seconds.
  text(n, seed)        English-like text: 4096-word Zipf vocabulary, sentences, ~80-col lines,
                       never 4 equal bytes in a row (RLE1 is the identity on it)
  mixed(n_segments)    1 MiB segments cycling text / random / runs / periodic / binary records /
                       2-state Markov / DNA-like / zeros
  adversarial()        dict of small adversarial inputs (runs, periodic, low entropy)
"""
import numpy as np

_LETTERS = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
_LETTER_P = np.array([12.7, 9.1, 8.2, 7.5, 7.0, 6.7, 6.3, 6.1, 6.0, 4.3, 4.0, 2.8, 2.8, 2.4, 2.4,
                      2.2, 2.0, 2.0, 1.9, 1.5, 1.0, 0.8, 0.15, 0.15, 0.1, 0.07])
_LETTER_P = _LETTER_P / _LETTER_P.sum()

TEXT900K_SEED = 0x00C0FFEE
MIXED256_SEED = 0x5EED0000
STREAM1G_SEED = 0x5EED1000
MIB = 1 << 20


def _vocab(rng, size=4096):
    lens = rng.integers(2, 11, size)
    mat = np.full((size, 12), ord(" "), dtype=np.uint8)
    letters = _LETTERS[rng.choice(26, size=(size, 10), p=_LETTER_P)]
    # no letter equal to its predecessor inside a word (so no 4-runs by construction)
    for k in range(1, 10):
        same = letters[:, k] == letters[:, k - 1]
        letters[same, k] = _LETTERS[(np.searchsorted(np.sort(_LETTERS), letters[same, k]) + 7) % 26]
        same = letters[:, k] == letters[:, k - 1]
        letters[same, k] = np.where(letters[same, k] == ord("e"), ord("a"), ord("e"))
    for k in range(10):
        m = lens > k
        mat[m, k] = letters[m, k]
    return mat, lens


def text(n, seed=TEXT900K_SEED):
    rng = np.random.default_rng(seed)
    mat, lens = _vocab(rng)
    p = 1.0 / np.arange(1, 4097)
    p /= p.sum()
    nwords = int(n / 5.0) + 64
    ids = rng.choice(4096, size=nwords, p=p)
    wl = lens[ids]
    # sentence ends every 8..20 words: word gets a trailing '.' (then the space)
    gaps = rng.integers(8, 21, nwords // 8 + 2)
    ends = np.cumsum(gaps)
    ends = ends[ends < nwords]
    is_end = np.zeros(nwords, dtype=bool)
    is_end[ends] = True
    tot = wl + 1 + is_end  # letters, optional '.', separator
    starts = np.concatenate(([0], np.cumsum(tot)[:-1]))
    total = int(starts[-1] + tot[-1])
    wid = np.repeat(np.arange(nwords), tot)
    pos = np.arange(total) - starts[wid]
    out = mat[ids[wid], np.minimum(pos, 11)]
    dot = is_end[wid] & (pos == wl[wid])
    out[dot] = ord(".")
    # capitalise the word after a sentence end (and the very first)
    cap = np.zeros(nwords, dtype=bool)
    cap[0] = True
    cap[ends[ends + 1 < nwords] + 1] = True
    first = cap[wid] & (pos == 0)
    out[first] -= 32
    # newline instead of the separator roughly every 80 columns
    sep = np.flatnonzero(pos == (tot[wid] - 1))
    col = sep // 80
    nl = sep[np.concatenate(([False], col[1:] != col[:-1]))]
    out[nl] = ord("\n")
    return out[:n].copy()


def _runs(rng, n):
    k = n // 20 + 64
    ln = rng.geometric(1.0 / 40.0, k)
    vals = rng.integers(0, 256, k, dtype=np.uint8)
    return np.repeat(vals, ln)[:n].copy()


def _periodic(rng, n):
    p = int(rng.integers(1, 65537))
    unit = rng.integers(0, 256, p, dtype=np.uint8)
    return np.tile(unit, n // p + 1)[:n].copy()


def _binary(rng, n):
    rec = n // 16 + 1
    a = np.zeros((rec, 16), dtype=np.uint8)
    ctr = (np.arange(rec, dtype=np.uint32) + rng.integers(0, 1 << 20)).astype("<u4")
    a[:, 0:4] = ctr.view(np.uint8).reshape(rec, 4)
    a[:, 4] = rng.integers(0, 4, rec)
    a[:, 5] = rng.choice(np.array([0, 1, 2, 255], dtype=np.uint8), rec, p=[0.7, 0.2, 0.05, 0.05])
    a[:, 6:8] = rng.integers(0, 16, (rec, 2))
    a[:, 8:12] = (rng.integers(0, 1000, rec).astype("<u4")).view(np.uint8).reshape(rec, 4)
    a[:, 12:16] = rng.integers(0, 256, (rec, 4)) * (rng.random((rec, 1)) < 0.1)
    return a.reshape(-1)[:n].copy()


def _markov2(rng, n):
    # two hidden states with geometric dwell times, each emitting a skewed 4-symbol law
    k = n // 50 + 64
    dwell = rng.geometric(1.0 / 100.0, k)
    state = np.repeat(np.arange(k) & 1, dwell)[:n]
    if state.size < n:
        state = np.concatenate((state, np.zeros(n - state.size, dtype=state.dtype)))
    u = rng.random(n)
    p0 = np.cumsum([0.85, 0.10, 0.04, 0.01])
    p1 = np.cumsum([0.05, 0.15, 0.30, 0.50])
    s0 = np.searchsorted(p0, u)
    s1 = np.searchsorted(p1, u)
    sym = np.where(state == 0, s0, s1)
    return (np.minimum(sym, 3) + ord("a")).astype(np.uint8)


def _dna(rng, n):
    # motif dictionary gives order-k like statistics (stand-in for an order-3 Markov chain)
    motifs = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (1024, 8))]
    p = 1.0 / np.arange(1, 1025) ** 0.7
    p /= p.sum()
    ids = rng.choice(1024, n // 8 + 1, p=p)
    return motifs[ids].reshape(-1)[:n].copy()


_KINDS = ["text", "random", "runs", "periodic", "binary", "markov", "dna", "zeros"]


def segment(kind, n, seed):
    rng = np.random.default_rng(seed)
    if kind == "text":
        return text(n, seed)
    if kind == "random":
        return rng.integers(0, 256, n, dtype=np.uint8)
    if kind == "runs":
        return _runs(rng, n)
    if kind == "periodic":
        return _periodic(rng, n)
    if kind == "binary":
        return _binary(rng, n)
    if kind == "markov":
        return _markov2(rng, n)
    if kind == "dna":
        return _dna(rng, n)
    if kind == "zeros":
        return np.zeros(n, dtype=np.uint8)
    raise ValueError(kind)


def mixed(n_segments=256, seed_base=MIXED256_SEED, seg_bytes=MIB):
    """n_segments x seg_bytes, kinds cycling (a)..(h)."""
    parts = [segment(_KINDS[i % 8], seg_bytes, seed_base + i) for i in range(n_segments)]
    return np.concatenate(parts)


def text900k():
    return text(899996, TEXT900K_SEED)


def _fib_word(n):
    a, b = b"a", b"ab"
    while len(b) < n:
        a, b = b, b + a
    return b[:n]


def _thue_morse(n):
    i = np.arange(n, dtype=np.uint64)
    x = i.copy()
    for s in (32, 16, 8, 4, 2, 1):
        x ^= x >> np.uint64(s)
    return ((x & np.uint64(1)).astype(np.uint8) + ord("a"))


_DEBRUIJN = {}


def _de_bruijn_bits(order=20):
    """Binary de Bruijn sequence B(2, order) by the prefer-one rule (2^order symbols)."""
    if order in _DEBRUIJN:
        return _DEBRUIJN[order]
    size = 1 << order
    mask = size - 1
    seen = bytearray(size)
    out = bytearray(size)
    w = 0
    seen[0] = 1
    k = 0
    # the sequence starts with `order` zeros (window 0 is taken); greedy: append 1 if new, else 0
    while k < size:
        w1 = ((w << 1) | 1) & mask
        if not seen[w1]:
            seen[w1] = 1
            w = w1
            out[k] = 1
        else:
            w0 = (w << 1) & mask
            if seen[w0]:
                break
            seen[w0] = 1
            w = w0
            out[k] = 0
        k += 1
    res = np.frombuffer(bytes(out[:k]), dtype=np.uint8).copy()
    _DEBRUIJN[order] = res
    return res


def square_block(level=9, seed=7):
    """Input whose first block at `level` is exactly w^2 (len(w) = (bs - 4) / 2, run-free), followed
    by unrelated data (SURVEY config 5: a block that is w^k with len(w) = 449 998, k = 2)."""
    rng = np.random.default_rng(seed)
    half = (level * 100000 - 4) // 2
    w = rng.integers(0, 255, half, dtype=np.uint8)
    # no byte four times in a row (RLE1 must be the identity) and no run across the seam
    for i in range(1, half):
        if w[i] == w[i - 1]:
            w[i] = (int(w[i]) + 1) % 255
    if w[0] == w[-1]:
        w[0] = (int(w[0]) + 2) % 255 if (int(w[0]) + 2) % 255 != w[1] else (int(w[0]) + 3) % 255
    return np.concatenate((w, w, rng.integers(0, 256, 5000, dtype=np.uint8)))


ADVERSARIAL_FAMILIES = ("zeros", "ff", "fb", "ab", "abc", "period255", "period256", "period257", "runlens",
                        "fibonacci", "thue_morse", "sawtooth", "sparse2", "square", "debruijn")


def adversarial_family(name, n, seed=99):
    """One adversarial input of n bytes."""
    import zlib
    rng = np.random.default_rng(seed + (zlib.crc32(name.encode()) & 0xffff))
    if name == "zeros":
        return np.zeros(n, dtype=np.uint8)
    if name == "ff":
        return np.full(n, 255, dtype=np.uint8)
    if name == "fb":
        return np.full(n, 0xFB, dtype=np.uint8)
    if name == "ab":
        return np.tile(np.frombuffer(b"ab", dtype=np.uint8), n // 2)
    if name == "abc":
        return np.tile(np.frombuffer(b"abc", dtype=np.uint8), n // 3)
    if name.startswith("period"):
        p = int(name[6:])
        return np.tile(rng.integers(0, 256, p, dtype=np.uint8), n // p + 1)[:n].copy()
    if name == "runlens":
        lens = np.array([3, 4, 5, 254, 255, 256, 259, 510])
        k = n // int(lens.mean()) + 8
        rl = lens[np.arange(k) % 8]
        vals = np.where(np.arange(k) & 1, ord("y"), ord("x")).astype(np.uint8)
        return np.repeat(vals, rl)[:n].copy()
    if name == "fibonacci":
        return np.frombuffer(_fib_word(n), dtype=np.uint8).copy()
    if name == "thue_morse":
        return _thue_morse(n)
    if name == "sawtooth":
        reps = n // 45000 + 2  # one period of the saw (k = 1..300) is 45 450 bytes
        kk = np.arange(1, 301)
        unit = np.repeat(np.tile(np.array([ord("a"), ord("b")], dtype=np.uint8), 300),
                         np.stack((kk, np.ones(300, dtype=np.int64)), axis=1).reshape(-1))
        return np.tile(unit, reps)[:n].copy()
    if name == "sparse2":
        return (rng.random(n) < 0.01).astype(np.uint8) + ord("0")
    if name == "square":
        half = rng.integers(0, 256, n // 2, dtype=np.uint8)
        return np.concatenate((half, half))
    if name == "debruijn":
        return (np.resize(_de_bruijn_bits(20), n) + ord("0")).astype(np.uint8)
    raise ValueError(name)


def adversarial(n=250000, seed=99):
    """Adversarial inputs at a size the CPU oracle handles in seconds."""
    rng = np.random.default_rng(seed)
    out = {}
    out["zeros"] = np.zeros(n, dtype=np.uint8)
    out["ff"] = np.full(n, 255, dtype=np.uint8)
    out["fb"] = np.full(n, 0xFB, dtype=np.uint8)
    out["ab"] = np.tile(np.frombuffer(b"ab", dtype=np.uint8), n // 2)
    out["abc"] = np.tile(np.frombuffer(b"abc", dtype=np.uint8), n // 3)
    for p in (255, 256, 257):
        out[f"period{p}"] = np.tile(rng.integers(0, 256, p, dtype=np.uint8), n // p + 1)[:n].copy()
    lens = np.array([3, 4, 5, 254, 255, 256, 259, 510])
    k = n // int(lens.mean()) + 8
    rl = lens[np.arange(k) % 8]
    vals = np.where(np.arange(k) & 1, ord("y"), ord("x")).astype(np.uint8)
    out["runlens"] = np.repeat(vals, rl)[:n].copy()
    out["fibonacci"] = np.frombuffer(_fib_word(n), dtype=np.uint8).copy()
    out["thue_morse"] = _thue_morse(n)
    saw = []
    kk = 1
    tot = 0
    while tot < n:
        saw.append(b"a" * kk + b"b")
        tot += kk + 1
        kk = kk % 300 + 1
    out["sawtooth"] = np.frombuffer(b"".join(saw), dtype=np.uint8)[:n].copy()
    out["sparse2"] = (rng.random(n) < 0.01).astype(np.uint8) + ord("0")
    half = rng.integers(0, 256, n // 2, dtype=np.uint8)
    out["square"] = np.concatenate((half, half))
    db = _de_bruijn_bits(20)
    out["debruijn"] = (np.resize(db, n) + ord("0")).astype(np.uint8)
    return out
