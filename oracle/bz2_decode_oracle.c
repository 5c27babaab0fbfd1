/* bz2_decode_oracle.c -- CPU restatement of the reference's DECODER.
 *
 * TEST INFRASTRUCTURE ONLY: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use it as the checker; the product path never calls it.
 *
 * Follows, statement by statement (32-bit Java int semantics, the "//#" Java-original lines where
 * the Swift text differs):
 *   S/BZip2BitInputStream.swift:53-146        bit reader (readBoolean / readUnary / readBits / readInteger)
 *   S/BZip2InputStream.swift:154-188,201-242  stream header, block / end marker dispatch, CRC fold
 *   S/BZip2BlockDecompressor.swift:199-258    symbol map, table count, selectors, delta-coded lengths
 *   S/BZip2BlockDecompressor.swift:266-321    Huffman symbols -> RUNA/RUNB runs + MTF -> BWT bytes
 *   S/BZip2BlockDecompressor.swift:328-355    merged-pointer inverse BWT set-up
 *   S/BZip2BlockDecompressor.swift:362-420    pointer walk + final run-length decoding + CRC
 *   S/BZip2HuffmanStageDecoder.swift:84-141   canonical decode tables; :149-181 nextSymbol
 *   S/BZip2MoveToFront.swift                   indexToFront
 *
 * Parity: a decoder is pinned by the format, not by the encoder that wrote the stream: every
 * correct decoder returns the same bytes.  tests/test_decode_oracle.py checks this restatement
 * against libbz2 (Python's bz2) on streams written by libbz2 and by our encoder oracle, and the
 * error paths by corrupting streams.  Not restated: randomised blocks (S/BZip2BlockDecompressor.swift:
 * 367-374 with its 512-entry RNUMS table, a pre-0.9.5 legacy no encoder has written since 1998):
 * ORC_D_RANDOMISED is returned for them.
 */
#include "bz2_oracle.h"
#include <stdlib.h>
#include <string.h>

enum {
    ORC_D_OK = 0,
    ORC_D_INSUFFICIENT = -1, /* "Insufficient data"                     BitInputStream :63,:92,:123 */
    ORC_D_HEADER = -2,       /* "Invalid BZip2 header"                  InputStream :177 */
    ORC_D_FORMAT = -3,       /* "BZip2 stream format error"             InputStream :240 */
    ORC_D_BLOCK_CRC = -4,    /* "BZip2 block CRC error"                 BlockDecompressor :456 */
    ORC_D_STREAM_CRC = -5,   /* "BZip2 stream CRC error"                InputStream :233 */
    ORC_D_TABLES = -6,       /* "BZip2 block Huffman tables invalid"    BlockDecompressor :235 */
    ORC_D_BLOCK_SIZE = -7,   /* "BZip2 block exceeds declared block size" :288,:307 */
    ORC_D_START_PTR = -8,    /* "BZip2 start pointer invalid"           :333 */
    ORC_D_DECODE = -9,       /* "Error decoding BZip2 block"            HuffmanStageDecoder :158,:179 */
    ORC_D_RANDOMISED = -10,  /* randomised block: not restated (see header) */
    ORC_D_CAPACITY = -11     /* output buffer too small (ours) */
};

typedef struct {
    const uint8_t *p;
    size_t n, pos;
    uint32_t buf; /* bitBuffer (Java int) */
    int cnt;      /* bitCount */
    int err;
} bitin_t;

static int in_byte(bitin_t *s) {
    if (s->pos >= s->n) {
        s->err = ORC_D_INSUFFICIENT;
        return 0;
    }
    return s->p[s->pos++];
}
/* S/BZip2BitInputStream.swift:114-137 */
static uint32_t read_bits(bitin_t *s, int count) {
    while (s->cnt < count) {
        if (s->err) return 0;
        s->buf = (s->buf << 8) | (uint32_t)in_byte(s);
        s->cnt += 8;
    }
    s->cnt -= count;
    return (s->buf >> s->cnt) & ((1u << count) - 1u);
}
/* :53-73 */
static int read_bool(bitin_t *s) {
    if (s->cnt > 0) s->cnt--;
    else {
        s->buf = (s->buf << 8) | (uint32_t)in_byte(s);
        s->cnt += 7;
    }
    return (s->buf & (1u << s->cnt)) != 0;
}
/* :80-106 */
static int read_unary(bitin_t *s) {
    int u = 0;
    for (;;) {
        if (s->err) return u;
        if (!read_bool(s)) return u;
        u++;
    }
}
/* :144-146 */
static uint32_t read_int(bitin_t *s) {
    uint32_t hi = read_bits(s, 16);
    return (hi << 16) | read_bits(s, 16);
}
static uint64_t bit_pos(const bitin_t *s) { return (uint64_t)s->pos * 8u - (uint64_t)s->cnt; }

#define D_MAXLEN 23 /* HUFFMAN_DECODE_MAXIMUM_CODE_LENGTH, S/BZip2Constants.swift:57 */
#define D_MAXSEL 18001

typedef struct {
    int min_len[6];
    int32_t bases[6][D_MAXLEN + 2];
    int32_t limits[6][D_MAXLEN + 1];
    int32_t symbols[6][258];
} dtables_t;

/* S/BZip2HuffmanStageDecoder.swift:84-141 */
static void make_tables(dtables_t *t, int alpha, int ntab, uint8_t lens[6][258]) {
    memset(t, 0, sizeof(*t));
    for (int tb = 0; tb < ntab; tb++) {
        int mn = D_MAXLEN, mx = 0;
        for (int i = 0; i < alpha; i++) {
            if (lens[tb][i] > mx) mx = lens[tb][i];
            if (lens[tb][i] < mn) mn = lens[tb][i];
        }
        t->min_len[tb] = mn;
        for (int i = 0; i < alpha; i++) t->bases[tb][lens[tb][i] + 1]++;
        for (int i = 1; i < D_MAXLEN + 2; i++) t->bases[tb][i] += t->bases[tb][i - 1];
        int32_t code = 0;
        for (int i = mn; i <= mx; i++) {
            int32_t base = code;
            code += t->bases[tb][i + 1] - t->bases[tb][i];
            t->bases[tb][i] = base - t->bases[tb][i];
            t->limits[tb][i] = code - 1;
            code <<= 1;
        }
        int ci = 0;
        for (int bl = mn; bl <= mx; bl++)
            for (int sy = 0; sy < alpha; sy++)
                if (lens[tb][sy] == bl) t->symbols[tb][ci++] = sy;
    }
}

/* One block, from just after its 48-bit magic to its end-of-block symbol: the BWT bytes.
 * S/BZip2BlockDecompressor.swift:467-483 (header), :199-258, :266-321.
 * bwt must hold block_size bytes. */
static int decode_block_to_bwt(bitin_t *s, int block_size, uint8_t *bwt, int *n_out, uint32_t *orig,
                               uint32_t *crc) {
    *crc = read_int(s);
    int randomised = read_bool(s);
    *orig = read_bits(s, 24);
    if (s->err) return s->err;
    if (randomised) return ORC_D_RANDOMISED;

    uint8_t symmap[256] = {0}; /* Array(repeating: 0, count: 256), S/BZip2BlockDecompressor.swift:119 */
    int nsym = 0;
    uint32_t ranges = read_bits(s, 16);
    for (int i = 0; i < 16; i++)
        if (ranges & ((1u << 15) >> i))
            for (int j = 0, k = i << 4; j < 16; j++, k++)
                if (read_bool(s)) symmap[nsym++] = (uint8_t)k;
    if (s->err) return s->err;
    const int eob = nsym + 1;
    int ntab = (int)read_bits(s, 3), nsel = (int)read_bits(s, 15);
    if (s->err) return s->err;
    if (ntab < 2 || ntab > 6 || nsel < 1 || nsel > D_MAXSEL) return ORC_D_TABLES;

    uint8_t *sel = (uint8_t *)malloc((size_t)nsel);
    uint8_t tmtf[256];
    for (int i = 0; i < 256; i++) tmtf[i] = (uint8_t)i;
    for (int i = 0; i < nsel; i++) {
        int u = read_unary(s);
        if (s->err) {
            free(sel);
            return s->err;
        }
        /* MoveToFront.indexToFront over the full 256-entry list: an index above the table count is
         * not rejected here (the reference does not either) */
        u &= 255;
        uint8_t v = tmtf[u];
        memmove(tmtf + 1, tmtf, (size_t)u);
        tmtf[0] = v;
        sel[i] = v;
    }
    static uint8_t lens[6][258]; /* not reentrant: the oracle decodes one stream at a time */
    memset(lens, 0, sizeof(lens));
    for (int t = 0; t < ntab; t++) {
        int cur = (int)read_bits(s, 5);
        for (int i = 0; i <= eob; i++) {
            while (read_bool(s)) {
                cur += read_bool(s) ? -1 : 1;
                if (s->err) break;
            }
            if (s->err) {
                free(sel);
                return s->err;
            }
            if (cur < 0 || cur > D_MAXLEN) { /* the reference indexes its tables out of range here */
                free(sel);
                return ORC_D_TABLES;
            }
            lens[t][i] = (uint8_t)cur;
        }
    }
    dtables_t *T = (dtables_t *)malloc(sizeof(dtables_t));
    make_tables(T, eob + 1, ntab, lens);

    uint8_t smtf[256];
    for (int i = 0; i < 256; i++) smtf[i] = (uint8_t)i;
    int len = 0, rep = 0, inc = 1, mtfv = 0, rc = ORC_D_OK;
    int gidx = -1, gpos = -1, cur = sel[0];
    for (;;) {
        /* nextSymbol, S/BZip2HuffmanStageDecoder.swift:149-181 */
        gpos++;
        if (gpos % 50 == 0) {
            gidx++;
            if (gidx == nsel) {
                rc = ORC_D_DECODE;
                break;
            }
            cur = sel[gidx];
        }
        if (cur >= 6) { /* a selector outside the tables: the reference would index out of range */
            rc = ORC_D_DECODE;
            break;
        }
        int cl = T->min_len[cur];
        int32_t bits = (int32_t)read_bits(s, cl);
        int sym = -1;
        while (cl <= D_MAXLEN) {
            if (bits <= T->limits[cur][cl]) {
                int32_t ix = bits - T->bases[cur][cl];
                if (ix < 0 || ix >= 258) break;
                sym = T->symbols[cur][ix];
                break;
            }
            bits = (bits << 1) | (int32_t)read_bits(s, 1);
            cl++;
        }
        if (s->err) {
            rc = s->err;
            break;
        }
        if (sym < 0) {
            rc = ORC_D_DECODE;
            break;
        }
        if (sym == 0) {
            rep += inc;
            inc <<= 1;
        } else if (sym == 1) {
            rep += inc << 1;
            inc <<= 1;
        } else {
            if (rep > 0) {
                if (len + rep > block_size) {
                    rc = ORC_D_BLOCK_SIZE;
                    break;
                }
                memset(bwt + len, symmap[mtfv], (size_t)rep);
                len += rep;
                rep = 0;
                inc = 1;
            }
            if (sym == eob) break;
            if (len >= block_size) {
                rc = ORC_D_BLOCK_SIZE;
                break;
            }
            int ix = (sym - 1) & 255;
            uint8_t v = smtf[ix];
            memmove(smtf + 1, smtf, (size_t)ix);
            smtf[0] = v;
            mtfv = v;
            bwt[len++] = symmap[mtfv];
        }
        if (rep > 2 * 900000 + 2) { /* keeps int arithmetic in range on hostile input */
            rc = ORC_D_BLOCK_SIZE;
            break;
        }
    }
    free(T);
    free(sel);
    *n_out = len;
    return rc;
}

/* S/BZip2BlockDecompressor.swift:328-355 and the walk of :362-379: tt[t] = t-th byte of the block
 * before the final run-length decoding. */
int orc_inverse_bwt(const uint8_t *bwt, int n, uint32_t orig, uint8_t *tt) {
    if (orig >= (uint32_t)n) return ORC_D_START_PTR;
    int32_t base[256] = {0}, cnt[256] = {0};
    for (int i = 0; i < n; i++) cnt[bwt[i]]++;
    for (int i = 1; i < 256; i++) base[i] = base[i - 1] + cnt[i - 1];
    uint32_t *merged = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
    for (int i = 0; i < n; i++) merged[base[bwt[i]]++] = ((uint32_t)i << 8) + bwt[i];
    uint32_t cur = merged[orig];
    for (int t = 0; t < n; t++) {
        tt[t] = (uint8_t)(cur & 0xff);
        cur = merged[cur >> 8];
    }
    free(merged);
    return ORC_D_OK;
}

/* S/BZip2BlockDecompressor.swift:386-420: final run-length decoding + CRC of the decoded bytes.
 * A block that ends where a repeat count is due is malformed (the reference would walk past its
 * data): ORC_D_FORMAT. */
int orc_unrle1(const uint8_t *tt, int n, uint8_t *out, size_t cap, size_t *out_len, uint32_t *crc) {
    size_t o = 0;
    int last = -1, acc = 0;
    for (int t = 0; t < n;) {
        int b = tt[t++];
        int rep = 1;
        if (b != last) {
            last = b;
            acc = 1;
        } else if (++acc == 4) {
            if (t == n) return ORC_D_FORMAT;
            rep = tt[t++] + 1;
            acc = 0;
        }
        if (o + (size_t)rep > cap) return ORC_D_CAPACITY;
        memset(out + o, b, (size_t)rep);
        o += (size_t)rep;
    }
    *out_len = o;
    *crc = ~orc_crc32_update(0xffffffffu, out, o); /* S/BZip2CRC32.swift:76-78 */
    return ORC_D_OK;
}

/* Stage entry for tests: decodes the block whose 48-bit magic starts at bit `bitpos`. */
int orc_decode_block(const uint8_t *in, size_t n, uint64_t bitpos, int block_size, uint8_t *bwt,
                     int *n_bwt, uint32_t *orig, uint32_t *crc, uint64_t *end_bit) {
    bitin_t s = {in, n, (size_t)(bitpos / 8), 0, 0, 0};
    if (bitpos % 8) read_bits(&s, (int)(bitpos % 8));
    uint32_t m1 = read_bits(&s, 24), m2 = read_bits(&s, 24);
    if (s.err) return s.err;
    if (m1 != 0x314159u || m2 != 0x265359u) return ORC_D_FORMAT;
    int rc = decode_block_to_bwt(&s, block_size, bwt, n_bwt, orig, crc);
    *end_bit = bit_pos(&s);
    return rc;
}

/* Whole stream, S/BZip2InputStream.swift:154-242.  *produced = bytes the reference would have
 * handed out before it threw (a block's bytes are out before its CRC is compared). */
int orc_decompress(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *produced,
                   int *n_blocks) {
    bitin_t s = {in, n, 0, 0, 0, 0};
    *produced = 0;
    *n_blocks = 0;
    uint32_t m1 = read_bits(&s, 16), m2 = read_bits(&s, 8);
    int level = (int)read_bits(&s, 8) - '0';
    if (s.err) return s.err;
    if (m1 != 0x425au || m2 != 0x68u || level < 1 || level > 9) return ORC_D_HEADER;
    const int block_size = level * 100000;
    uint8_t *bwt = (uint8_t *)malloc((size_t)block_size), *tt = (uint8_t *)malloc((size_t)block_size);
    uint32_t stream_crc = 0;
    int rc = ORC_D_OK;
    for (;;) {
        uint32_t a = read_bits(&s, 24), b = read_bits(&s, 24);
        if (s.err) {
            rc = s.err;
            break;
        }
        if (a == 0x314159u && b == 0x265359u) {
            int nb = 0;
            uint32_t orig = 0, crc = 0, got = 0;
            rc = decode_block_to_bwt(&s, block_size, bwt, &nb, &orig, &crc);
            if (rc) break;
            rc = orc_inverse_bwt(bwt, nb, orig, tt);
            if (rc) break;
            size_t ol = 0;
            rc = orc_unrle1(tt, nb, out + *produced, cap - *produced, &ol, &got);
            if (rc) break;
            *produced += ol;
            (*n_blocks)++;
            if (got != crc) {
                rc = ORC_D_BLOCK_CRC;
                break;
            }
            stream_crc = ((stream_crc << 1) | (stream_crc >> 31)) ^ crc;
        } else if (a == 0x177245u && b == 0x385090u) {
            uint32_t stored = read_int(&s);
            if (s.err) rc = s.err;
            else if (stored != stream_crc) rc = ORC_D_STREAM_CRC;
            break;
        } else {
            rc = ORC_D_FORMAT;
            break;
        }
    }
    free(bwt);
    free(tt);
    return rc;
}
