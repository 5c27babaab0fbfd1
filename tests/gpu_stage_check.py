"""Ad-hoc GPU bring-up script (not collected by pytest): stage-by-stage parity vs the oracle."""
import sys, os, time, bz2
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import orc
from bzip2_amd import Context

rng = np.random.default_rng(7)
def text(n):
    words = [bytes(rng.integers(97, 123, rng.integers(2, 10), dtype=np.uint8)) for _ in range(500)]
    out = bytearray()
    while len(out) < n:
        out += words[int(rng.zipf(1.3)) % 500] + b" "
    return bytes(out[:n])

cases = {
    "one": b"x", "two": b"ab", "aaaa": b"aaaaa", "small": b"banana_bandana",
    "rand4k": rng.integers(0, 256, 4000, dtype=np.uint8).tobytes(),
    "text50k": text(50000), "abab": b"ab" * 3000, "abc": b"abc" * 2000,
    "lowent": rng.integers(0, 2, 30000, dtype=np.uint8).tobytes(),
    "zeros": bytes(20000), "runs": b"".join(bytes([rng.integers(0, 256)]) * int(rng.geometric(1 / 40)) for _ in range(800)),
    "text300k": text(300000),
}
ctx = Context()
ok_all = True
def check(name, cond):
    global ok_all
    print(("PASS " if cond else "FAIL ") + name, flush=True)
    ok_all &= bool(cond)

# oracle stages
blocks, presents, crcs = [], [], []
names = []
for k, d in cases.items():
    pos = 0
    while pos < len(d):
        used, blk, pres, crc = orc.rle1_block(d[pos:], 900000)
        pos += used
        blocks.append(blk); presents.append(pres); crcs.append(crc); names.append(k)
t = time.time()
res = ctx.bwt(blocks, 900000)
print("bwt time", time.time() - t, flush=True)
obw = []
for nm, blk, (bw, orig, rounds) in zip(names, blocks, res):
    ob, oo, per = orc.bwt(blk, 900000)
    obw.append(ob)
    check(f"bwt {nm} n={len(blk)} rounds={rounds} per={per}", np.array_equal(ob, bw) and oo == orig)
res = ctx.mtf_rle2(obw, presents)
om = []
for nm, bw, pres, (mtf, alpha, freq) in zip(names, obw, presents, res):
    m, a, f = orc.mtf_rle2(bw, pres)
    om.append((m, a, f))
    check(f"mtf {nm} len={len(m)}", np.array_equal(m, mtf) and a == alpha and np.array_equal(f, freq))
res = ctx.huffman([x[0] for x in om], [x[1] for x in om], [x[2] for x in om])
for nm, (m, a, f), (nt, lens, sel) in zip(names, om, res):
    ont, ol, osel = orc.huffman_optimise(m, a, f)
    check(f"huff {nm} nt={nt}", nt == ont and np.array_equal(ol[:nt, :a], lens[:nt, :a]) and np.array_equal(osel, sel))
res = ctx.block_compress(blocks, presents, crcs, 900000)
for nm, blk, pres, crc, (bits, nbits) in zip(names, blocks, presents, crcs, res):
    ob, on = orc.block_compress(blk, pres, crc, 900000)
    check(f"block {nm} nbits={nbits}", on == nbits and np.array_equal(ob, bits))
for k, d in cases.items():
    for lvl in (1, 9):
        o, nb = orc.compress(d, lvl)
        g = ctx.compress(d, lvl)
        check(f"stream {k} L{lvl} blocks={nb} len={len(g)}", g == o and bz2.decompress(g) == d)
o, _ = orc.compress(b"", 9)
check("empty", ctx.compress(b"", 9) == o)
print("ALL OK" if ok_all else "SOME FAILED")
sys.exit(0 if ok_all else 1)
