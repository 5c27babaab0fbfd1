/*
 * bz2_oracle.h -- CPU restatement of the bastie/bzip2 (jbzip2-derived) block compressor.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / the timed CPU baseline.  The product (bzip2_amd/csrc) never links it.
 *
 * PARITY UNPINNED: the reference (Swift port of jbzip2) ships no runnable test, golden
 * vector or fixture (Tests/bzip2JavApiTests/bzip2JavApiTests.swift:4-6 is empty) and cannot
 * be built here (no Swift/Java toolchain, remote JavApi4Swift dependency).  This restatement
 * follows the Java-original statements preserved in the Swift sources (32-bit int
 * semantics) and is pinned only by format checks: every stream must decode, with CRC
 * verification, through libbz2 / Python bz2 / bzip2 1.0.8, and the BWT is cross-checked
 * against a brute-force rotation sort.
 *
 * All citations are into /root/reference/Sources/bzip2JavApi/ ("S/").
 */
#ifndef BZ2_ORACLE_H
#define BZ2_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_ALPHA 258
#define ORC_MAX_TABLES 6
#define ORC_GROUP 50
#define ORC_MAX_SELECTORS 18002

/* S/BZip2CRC32.swift:84-105 -- raw register update (init 0xffffffff, caller inverts). */
uint32_t orc_crc32_update(uint32_t crc, const uint8_t *p, size_t n);

/* S/BZip2BlockCompressor.swift:198-252,151-191,258-266 -- RLE1 one block.
 * Consumes bytes from in[0..n_in) until the block refuses one (or input ends), flushes the
 * pending run.  block must hold block_size+1 bytes.  Returns bytes consumed. */
size_t orc_rle1_block(const uint8_t *in, size_t n_in, int block_size, uint8_t *block,
                      int *block_len, uint8_t present[256], uint32_t *block_crc);

/* S/BZip2DivSufSort.swift:3487-3506 contract: last column of sorted cyclic rotations and
 * the sorted position of rotation 0.  *periodic is set to the period count k (block == w^k;
 * 1 if aperiodic).  The k rotations equal to rotation 0 are indistinguishable, so the origin
 * pointer of a periodic block depends on the sorter; tie_rule selects it:
 *   0  tier 2: the reference's own sorter restated (divsufsort_restate.c)
 *   1  tier 1: prefix doubling, first position of the class
 *   2  tier 1: prefix doubling, last position of the class
 * Tier 1 is an independent algorithm kept as a cross-check: same bytes always, same origin
 * pointer on every aperiodic block.
 * declared_block_size = the size the reference's caller declared (level * 100000,
 * S/BZip2OutputStream.swift:177): it is the length of the zero-filled int array the sorter works
 * in (S/BZip2BlockCompressor.swift:318), and tier 2's pointer on a periodic block can depend on it
 * (one read lands in that zero tail, S/BZip2DivSufSort.swift:2589-2592).  0 = the smallest
 * multiple of 100000 that holds n.  Both are arguments, never process state. */
void orc_bwt(const uint8_t *T, int n, int declared_block_size, int tie_rule, uint8_t *bwt, int *orig,
             int *periodic);
int orc_declared_block_size(int n, int declared_block_size);

/* Tier 2: restatement of the reference's BZip2DivSufSort (divsufsort_restate.c).  T must have
 * room for n + 1 bytes (T[n] := T[0]).  It is what orc_bwt uses under tie rule 0. */
void orc_bwt_divsufsort(uint8_t *T, int n, int declared_block_size, uint8_t *bwt, int *orig);
int orc_divsufsort_bwt(const uint8_t *T, int *sa_out, int n, int sa_len, int *trap);
/* Number of blocks so far on which the reference would have failed with an index error. */
int orc_divsufsort_traps(void);
/* Brute-force cross-check (n small): sorts rotations with memcmp on the doubled string. */
void orc_bwt_bruteforce(const uint8_t *T, int n, uint8_t *bwt, int *orig_first, int *orig_last);

/* BZip2MTFAndRLE2StageEncoder.java:50-134 */
void orc_mtf_rle2(const uint8_t *bwt, int n, const uint8_t present[256], uint16_t *mtf,
                  int *mtf_len, int *alpha, int32_t freq[ORC_MAX_ALPHA]);

/* S/BZip2HuffmanAllocator.swift:197-223 (Java-original loops) */
void orc_huffman_allocate(int32_t *a, int len, int max_len);

/* S/BZip2HuffmanStageEncoder.swift:107-135 */
void orc_huffman_code_lengths(int alpha, const int32_t *freq, int32_t *lengths);

/* S/BZip2HuffmanStageEncoder.swift:82-98,144-181,192-248,360-367: table count, seeds and the
 * four optimisation passes.  lengths is [ORC_MAX_TABLES][ORC_MAX_ALPHA]; selectors holds
 * ceil(mtf_len/50) entries.  If iter_lengths != NULL it receives the lengths after each of the
 * 4 passes ([4][6][258]).  Returns table count. */
int orc_huffman_optimise(const uint16_t *mtf, int mtf_len, int alpha, const int32_t *freq,
                         int32_t lengths[ORC_MAX_TABLES][ORC_MAX_ALPHA], uint8_t *selectors,
                         int *n_selectors, int32_t *iter_lengths);

/* S/BZip2HuffmanStageEncoder.swift:254-287 */
void orc_huffman_codes(int n_tables, int alpha, int32_t lengths[ORC_MAX_TABLES][ORC_MAX_ALPHA],
                       int32_t merged[ORC_MAX_TABLES][ORC_MAX_ALPHA]);

/* One whole block, starting at bit 0 of `bits` (MSB first), given the RLE1 output.
 * S/BZip2BlockCompressor.swift:258-289.  Returns number of bits written.  `bits` must be
 * zeroed by the caller and hold at least n*5/4+2048 bytes... use orc_block_bound(n). */
uint64_t orc_block_compress(const uint8_t *rle_block, int n, const uint8_t present[256],
                            uint32_t block_crc, int declared_block_size, int tie_rule, uint8_t *bits);
size_t orc_block_bound(int n);

/* Whole stream: S/BZip2OutputStream.swift:85-104,131-161,171-189.  Returns bytes written,
 * or 0 if cap is too small / level invalid.  n_blocks (optional) receives the block count. */
size_t orc_compress(const uint8_t *in, size_t n, int level, int tie_rule, uint8_t *out, size_t cap,
                    int *n_blocks);
size_t orc_compress_bound(size_t n);

/* Same, but block-parallel over `threads` host threads (CPU baseline, all cores). */
size_t orc_compress_mt(const uint8_t *in, size_t n, int level, int tie_rule, uint8_t *out, size_t cap,
                       int threads, int *n_blocks);

/* Block boundaries only: fills starts[0..max) with input offsets where blocks begin,
 * returns the number of blocks. */
int orc_split(const uint8_t *in, size_t n, int level, uint64_t *starts, int max);

#ifdef __cplusplus
}
#endif
/* ---- decoder restatement (bz2_decode_oracle.c).  Return 0 or a negative ORC_D_* code. */
int orc_decompress(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *produced,
                   int *n_blocks);
int orc_decode_block(const uint8_t *in, size_t n, uint64_t bitpos, int block_size, uint8_t *bwt,
                     int *n_bwt, uint32_t *orig, uint32_t *crc, uint64_t *end_bit);
int orc_inverse_bwt(const uint8_t *bwt, int n, uint32_t orig, uint8_t *tt);
int orc_unrle1(const uint8_t *tt, int n, uint8_t *out, size_t cap, size_t *out_len, uint32_t *crc);

#endif
