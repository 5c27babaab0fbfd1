/*
 * bz2hip.h -- C ABI of libbz2hip: the MI355X (gfx950) bzip2 block compressor.
 *
 * This is the drop-in boundary for the hot path of bastie/bzip2 (Swift port of jbzip2):
 * everything below BZip2OutputStream.closeBlock() plus the RLE1/CRC ingest loop.  The
 * reference has no FFI of its own (pure Swift); each entry point cites the reference
 * interface it replaces ("S/" = Sources/bzip2JavApi/).  Plain pointers and sizes only.
 *
 * Conventions: every function returns BZ2HIP_OK (0) or a negative error code and never
 * throws; bz2hip_last_error() gives the message.  A context is bound to one GPU and is not
 * thread-safe (the reference streams are not either: S/BZip2OutputStream.swift:29).
 * Pointers named d_* are device (HBM) pointers on the context's GPU; all others are host.
 */
#ifndef BZ2HIP_H
#define BZ2HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BZ2HIP_OK 0
#define BZ2HIP_E_PARAM (-1)   /* bad argument (level outside 1..9, null pointer, ...) */
#define BZ2HIP_E_NOMEM (-2)   /* device or host allocation failed */
#define BZ2HIP_E_HIP (-3)     /* a HIP runtime call failed */
#define BZ2HIP_E_CAP (-4)     /* caller's output buffer too small */
#define BZ2HIP_E_STATE (-5)   /* call sequence error, e.g. write after finish */
/* decoder: the reference's exceptions, in the order it would throw them */
#define BZ2HIP_E_INSUFFICIENT (-6) /* "Insufficient data"            S/BZip2BitInputStream.swift:63,92,123 */
#define BZ2HIP_E_HEADER (-7)       /* "Invalid BZip2 header"         S/BZip2InputStream.swift:177 */
#define BZ2HIP_E_FORMAT (-8)       /* "BZip2 stream format error"    S/BZip2InputStream.swift:240 */
#define BZ2HIP_E_BLOCK_CRC (-9)    /* "BZip2 block CRC error"        S/BZip2BlockDecompressor.swift:456 */
#define BZ2HIP_E_STREAM_CRC (-10)  /* "BZip2 stream CRC error"       S/BZip2InputStream.swift:233 */
#define BZ2HIP_E_TABLES (-11)      /* "BZip2 block Huffman tables invalid"       S/BZip2BlockDecompressor.swift:235 */
#define BZ2HIP_E_BLOCK_SIZE (-12)  /* "BZip2 block exceeds declared block size"  :288,:307 */
#define BZ2HIP_E_START_PTR (-13)   /* "BZip2 start pointer invalid"              :333 */
#define BZ2HIP_E_DECODE (-14)      /* "Error decoding BZip2 block"   S/BZip2HuffmanStageDecoder.swift:158,179 */
#define BZ2HIP_E_RANDOMISED (-15)  /* randomised block (S/BZip2BlockDecompressor.swift:367-374): not supported */

#define BZ2HIP_MAX_ALPHA 258
#define BZ2HIP_MAX_TABLES 6

typedef struct bz2hip_ctx bz2hip_ctx;

/* A block that is a perfect power w^k has k rotations equal to rotation 0; any of their sorted
 * positions decodes correctly, the reference emits the one its sorter happens to produce
 * (S/BZip2DivSufSort.swift:3472-3477).  REFERENCE replays that sorter on the device for such
 * blocks (one lane per block, slow but exact); FIRST/LAST skip it. */
#define BZ2HIP_TIE_REFERENCE 0
#define BZ2HIP_TIE_FIRST 1
#define BZ2HIP_TIE_LAST 2

#define BZ2HIP_MAX_DEVICES 8

typedef struct bz2hip_opts {
    int device;            /* HIP device ordinal; -1 = current device (used when n_devices <= 1) */
    int max_batch_blocks;  /* blocks sorted per batch (0 = default 1024) */
    int tie_rule;          /* origin pointer of periodic blocks (block == w^k): BZ2HIP_TIE_* */
    int profile;           /* HIP-event timing per kernel class (bz2hip_profile_*): 1 = the
                              bandwidth-bound kernels only (cheap), 2 = every class */
    int n_devices;         /* 0 or 1: one GPU (`device`).  2..8: the host-buffer entry points
                              (bz2hip_compress, bz2hip_stream_*) shard every stream over
                              devices[0..n_devices): one worker thread and one block queue per GPU,
                              ordered concatenation on the calling thread (the reference has one
                              writer per stream, S/BZip2OutputStream.swift:29).  An ordinal may
                              appear more than once (two workers on one GPU).  Device-pointer and
                              stage entry points run on devices[0]. */
    int devices[BZ2HIP_MAX_DEVICES];
} bz2hip_opts;

/* Context life cycle (no counterpart in the reference; holds the GPU workspace). */
int bz2hip_create(bz2hip_ctx **ctx, const bz2hip_opts *opts);
void bz2hip_destroy(bz2hip_ctx *ctx);
const char *bz2hip_last_error(const bz2hip_ctx *ctx);
const char *bz2hip_version(void);

/* Page-locked host memory for the buffers handed to the host-pointer entry points: transfers from
 * and to such buffers run at the full PCIe rate and asynchronously (pageable memory is staged by
 * the runtime at roughly half the rate).  A binding that owns its buffers -- as the reference's
 * BZip2BlockCompressor owns `block` (S/BZip2BlockCompressor.swift:317) -- allocates them here. */
int bz2hip_host_alloc(void **p, size_t bytes);
void bz2hip_host_free(void *p);

/* Upper bound of the compressed size of n input bytes (stream header + trailer included). */
size_t bz2hip_compress_bound(size_t n);

/* ---- whole-stream entry points -------------------------------------------------------
 * Replace BZip2OutputStream(out, level); write(data, 0, n); close()
 * (S/BZip2OutputStream.swift:85-104,110-113,149-161,171-189).  Output is the complete
 * .bz2 stream ("BZh" + level, blocks, end marker, stream CRC, zero pad). */
int bz2hip_compress(bz2hip_ctx *ctx, const uint8_t *in, size_t n, int level, uint8_t *out,
                    size_t cap, size_t *out_len);
/* Same with input and output resident in HBM (no PCIe transfer). */
int bz2hip_compress_device(bz2hip_ctx *ctx, const void *d_in, size_t n, int level, void *d_out,
                           size_t cap, size_t *out_len);

/* Blocks only, for sharding one stream over several GPUs: compresses d_in[0..n) -- which must
 * begin at a block boundary of the whole stream and end at one (or at the end of the data) --
 * into bit-abutted blocks starting at bit 0 of d_out, without stream header or trailer.
 * nbits = total bit length; crcs[b] = block CRCs in order (what closeBlock folds into the
 * stream CRC, S/BZip2OutputStream.swift:131-141).  The host stitches shards in order
 * (bzip2_amd/parallel.py). */
int bz2hip_compress_blocks_device(bz2hip_ctx *ctx, const void *d_in, size_t n, int level,
                                  void *d_out, size_t cap, uint64_t *nbits, uint32_t *crcs,
                                  uint32_t max_blocks, uint32_t *n_blocks);

/* Host-pointer form of the above (stages through the context's device buffers). */
int bz2hip_compress_blocks(bz2hip_ctx *ctx, const uint8_t *in, size_t n, int level, uint8_t *out,
                           size_t cap, uint64_t *nbits, uint32_t *crcs, uint32_t max_blocks,
                           uint32_t *n_blocks);

/* Allocation-free variant of bz2hip_compress_device.  It has NO counterpart in the reference: it
 * exists for callers that cannot allocate device memory -- tight serving loops, harnesses that
 * forbid hipMalloc.  It makes no device allocation but is NOT capturable into a HIP graph: like
 * bz2hip_compress_device it makes synchronous read-backs (block chain, round control, bit
 * cursor).  The whole workspace is carved from a static
 * device array of BZ2HIP_NOALLOC_SCRATCH_MIB MiB that every GPU the library runs on holds until
 * the process ends; the call makes no hipMalloc / hipFree.  n must not exceed
 * BZ2HIP_NOALLOC_MAX_INPUT (BZ2HIP_E_PARAM otherwise); output is byte-identical to
 * bz2hip_compress_device.  All calls on one GPU share that GPU's array: calls that overlap on one
 * GPU (two contexts or two threads) race. */
#define BZ2HIP_NOALLOC_SCRATCH_MIB 64
#define BZ2HIP_NOALLOC_MAX_INPUT (1u << 20)
int bz2hip_compress_device_noalloc(bz2hip_ctx *ctx, const void *d_in, size_t n, int level,
                                   void *d_out, size_t cap, size_t *out_len);

/* ---- one stream sharded over several GPUs / processes -----------------------------------
 * No counterpart in the reference (single-threaded, S/BZip2OutputStream.swift:29); the block
 * boundaries are those of BZip2BlockCompressor.write (S/BZip2BlockCompressor.swift:198-227), and
 * what the shards produce is concatenated as closeBlock / finish would (S/BZip2OutputStream.swift:
 * 131-141,149-161).  A shard is a slice d_in[0..avail) of the stream: the bytes of its nominal
 * range [.., limit) plus lookahead, because its last block may end beyond `limit`.
 *   bz2hip_shard_scan      run-length scan of the slice (asynchronous; does not depend on where
 *                          blocks start, so every shard scans at once); avail <= 1 GiB
 *   bz2hip_shard_chain     the serial block chain: from `first` (a block start, slice-relative;
 *                          0 for the first shard, the previous shard's *next otherwise) it takes
 *                          the blocks that start below `limit`.  *next = slice-relative start of
 *                          the first block NOT taken = where the next shard continues.  final != 0:
 *                          the stream ends at d_in + avail.  A last block that runs into the end
 *                          of the slice without `final` is not taken (*next < limit then: scan
 *                          again with more lookahead).
 *   bz2hip_shard_compress  compresses the blocks of the last chain into bit-abutted blocks from
 *                          bit 0 of d_out (no stream header / trailer), like
 *                          bz2hip_compress_blocks_device. */
int bz2hip_shard_scan(bz2hip_ctx *ctx, const void *d_in, size_t avail, int level);
int bz2hip_shard_chain(bz2hip_ctx *ctx, uint64_t first, uint64_t limit, int final,
                       uint32_t *n_blocks, uint64_t *next);
int bz2hip_shard_compress(bz2hip_ctx *ctx, void *d_out, size_t cap, uint64_t *nbits,
                          uint32_t *crcs, uint32_t max_blocks, uint32_t *n_blocks);

/* bz2hip_stitch with the segments and the stream resident in HBM (device pointers in the host
 * array d_seg_bits, 4-byte aligned; CRCs on the host): K8 across shards on one GPU. */
int bz2hip_stitch_device(bz2hip_ctx *ctx, int level, uint32_t nseg, const void *const *d_seg_bits,
                         const uint64_t *seg_nbits, const uint32_t *const *seg_crcs,
                         const uint32_t *seg_ncrc, void *d_out, size_t cap, size_t *out_len);

/* Ordered concatenation of block segments (from bz2hip_compress_blocks*) into one stream: the
 * sequential part of BZip2OutputStream -- header :181-184, bit-abutted blocks, CRC fold :139,
 * end marker + stream CRC + zero pad :154-157 (S/BZip2OutputStream.swift).  Host only. */
int bz2hip_stitch(int level, uint32_t nseg, const uint8_t *const *seg_bits,
                  const uint64_t *seg_nbits, const uint32_t *const *seg_crcs,
                  const uint32_t *seg_ncrc, uint8_t *out, size_t cap, size_t *out_len);

/* ---- streaming entry points ----------------------------------------------------------
 * bz2hip_stream_begin  = BZip2OutputStream.init        S/BZip2OutputStream.swift:171-189
 * bz2hip_stream_write  = write([UInt8], off, len)      :85-104 (buffers; blocks are cut by the
 *                        RLE1 acceptance rule of BZip2BlockCompressor.write :198-227)
 * bz2hip_stream_finish = finish()                      :149-161
 * bz2hip_stream_read   drains compressed bytes in stream order (the bytes the reference would
 *                        have passed to java.io.OutputStream.write, S/BZip2BitOutputStream.swift:58,93)
 * Writing after finish returns BZ2HIP_E_STATE ("Write beyond end of stream", :69-71,89-91).
 * On a single-device context the stream is asynchronous, as the reference's write() is for its
 * caller (it returns once the bytes are taken, blocks are closed later): write copies into one
 * of two page-locked buffers (2 x 64 MiB each) and a worker thread of the context uploads,
 * compresses and queues the output while the caller fills the other buffer; a buffer is handed
 * over when it is full, or when the worker is idle and 32 MiB have gathered.  Compressed bytes
 * therefore become readable some time after the write that completed their block; after
 * bz2hip_stream_finish has returned all of them are.  An error of the worker is reported by the
 * next write / finish.  Any other call on the context while a stream is open first waits for the
 * worker and leaves the stream intact.  Contexts with a device list compress inside write / finish. */
int bz2hip_stream_begin(bz2hip_ctx *ctx, int level);
int bz2hip_stream_write(bz2hip_ctx *ctx, const uint8_t *p, size_t n);
int bz2hip_stream_finish(bz2hip_ctx *ctx);
int bz2hip_stream_read(bz2hip_ctx *ctx, uint8_t *out, size_t cap, size_t *produced);
size_t bz2hip_stream_pending(const bz2hip_ctx *ctx);

/* ---- block / stage entry points (host pointers; batched over `count` blocks) ---------
 * Blocks are laid out at `stride` bytes (or elements) from each other. */

/* Block boundaries + RLE1 + CRC: BZip2BlockCompressor.write/writeRun
 * (S/BZip2BlockCompressor.swift:151-252) and BZip2CRC32 (S/BZip2CRC32.swift:76-105).
 * starts[b] = input offset of block b; rle (stride level*100000+8) receives the RLE1 blocks
 * (rle .. crc may all be NULL: boundaries only).  Any input length: the input is scanned in
 * windows of 1 GiB. */
int bz2hip_rle1_split(bz2hip_ctx *ctx, const uint8_t *in, size_t n, int level,
                      uint32_t max_blocks, uint32_t *n_blocks, uint64_t *starts,
                      uint8_t *rle, uint32_t *rle_len, uint8_t *present /*[b][256]*/,
                      uint32_t *crc);

/* BZip2DivSufSort(T, SA, n).bwt()  S/BZip2DivSufSort.swift:3487-3518: last column of the sorted
 * cyclic rotations and the origin pointer.  rounds (optional) = doubling rounds executed.
 * declared_block_size = the block size the caller's SA array was sized for (blockSize + 1 ints,
 * S/BZip2BlockCompressor.swift:318; level * 100000): the origin pointer of a periodic block
 * depends on it.  0 = the smallest level that holds the longest block. */
int bz2hip_bwt(bz2hip_ctx *ctx, const uint8_t *blocks, const uint32_t *n, uint32_t count,
               uint32_t stride, uint8_t *bwt, uint32_t *orig, uint32_t *rounds,
               uint32_t declared_block_size);

/* BZip2MTFAndRLE2StageEncoder(bwt, n, present).encode() + getters
 * (S/BZip2MTFAndRLE2StageEncoder.swift:43-163).  mtf has stride+2 elements per block. */
int bz2hip_mtf_rle2(bz2hip_ctx *ctx, const uint8_t *bwt, const uint32_t *n, uint32_t count,
                    uint32_t stride, const uint8_t *present, uint16_t *mtf, uint32_t *mtf_len,
                    uint32_t *alpha, int32_t *freq /*[b][258]*/);

/* BZip2HuffmanStageEncoder: selectTableCount, generateHuffmanOptimisationSeeds, 4x
 * optimiseSelectorsAndHuffmanTables (S/BZip2HuffmanStageEncoder.swift:82-98,144-248,360-366)
 * incl. HuffmanAllocator.allocateHuffmanCodeLengths (S/BZip2HuffmanAllocator.swift:197-223).
 * lengths is [b][6][258] (int32), selectors has sel_stride bytes per block. */
int bz2hip_huffman(bz2hip_ctx *ctx, const uint16_t *mtf, const uint32_t *mtf_len,
                   const uint32_t *alpha, const int32_t *freq, uint32_t count, uint32_t stride,
                   uint32_t *n_tables, int32_t *lengths, uint8_t *selectors,
                   uint32_t sel_stride, uint32_t *n_selectors);

/* BZip2BlockCompressor.close() S/BZip2BlockCompressor.swift:258-289 for `count` RLE1'd blocks:
 * BWT, header, symbol map, MTF/RLE2, Huffman, bit packing.  Each block's bits start at bit 0
 * (MSB first) of bits + b*bits_stride; nbits[b] = exact bit length. */
int bz2hip_block_compress(bz2hip_ctx *ctx, const uint8_t *rle_blocks, const uint32_t *n,
                          uint32_t count, uint32_t stride, const uint8_t *present,
                          const uint32_t *crc, uint8_t *bits, size_t bits_stride,
                          uint64_t *nbits, uint32_t declared_block_size /* as in bz2hip_bwt */);

/* ---- Decoder.  Replaces BZip2InputStream.read(...) over a whole stream
 * (S/BZip2InputStream.swift:91-128,154-242) and the classes under it (BZip2BlockDecompressor,
 * BZip2HuffmanStageDecoder, BZip2BitInputStream): all blocks are decoded at once on the GPU.
 * in = one complete .bz2 stream ("BZh" + level ...; bytes after the end marker are ignored as the
 * reference ignores them).  Returns BZ2HIP_OK or the first error the reference would throw;
 * *produced = the bytes it would have handed out before that, written to out (a block's bytes are
 * out before its CRC is compared).  BZ2HIP_E_CAP: a block did not fit in cap (*produced = the blocks
 * before it).  The _device form takes device pointers. */
int bz2hip_decompress(bz2hip_ctx *ctx, const uint8_t *in, size_t n, uint8_t *out, size_t cap,
                      size_t *produced);
int bz2hip_decompress_device(bz2hip_ctx *ctx, const void *d_in, size_t n, void *d_out, size_t cap,
                             size_t *produced);

/* Input bytes the stream of the last bz2hip_decompress* call occupied (header through stream CRC and
 * its zero padding), 0 unless it returned BZ2HIP_OK.  The reference's decoder stops at the first
 * end marker (S/BZip2InputStream.swift:201-242); a front end that wants the bzip2 tool's behaviour
 * for concatenated streams calls again at in + consumed, or uses bz2hip_decompress_streams.  After a
 * _streams call: the bytes through the last stream that was decoded completely, whatever it returned. */
int bz2hip_decode_consumed(bz2hip_ctx *ctx, size_t *consumed);

/* Concatenated streams in one call: what a caller gets who constructs one BZip2InputStream
 * (S/BZip2InputStream.swift:154-242) after the other over the same input until it is used up, with
 * the outputs joined (the reference has no such loop of its own; the bzip2 tool and pbzip2-style
 * writers define the format).  After a stream's end marker the call goes on with the stream that
 * follows if its header names the SAME level as the first; it stops, with BZ2HIP_OK, at the end of the
 * input or at bytes that are no such header (bz2hip_decode_consumed tells where: a caller continues
 * there for a stream of another level).  The blocks of all streams are found by one scan and decoded
 * in the same batches, so thousands of small streams fill the GPU like one large stream.
 * An error is the one the failing stream's BZip2InputStream would throw; *produced then counts every
 * earlier stream completely plus that stream's bytes before the error.  *n_streams = streams decoded
 * completely. */
int bz2hip_decompress_streams(bz2hip_ctx *ctx, const uint8_t *in, size_t n, uint8_t *out, size_t cap,
                              size_t *produced, uint32_t *n_streams);
int bz2hip_decompress_streams_device(bz2hip_ctx *ctx, const void *d_in, size_t n, void *d_out,
                                     size_t cap, size_t *produced, uint32_t *n_streams);

/* ---- profiling (profile=1): accumulated HIP-event time per kernel class --------------- */
#define BZ2HIP_PROF_MAX 48
int bz2hip_profile_reset(bz2hip_ctx *ctx);
/* names[i], ms[i], launches[i], bytes[i] (algorithmic bytes) for i < *count */
int bz2hip_profile_get(bz2hip_ctx *ctx, int *count, const char **names, double *ms,
                       uint64_t *launches, double *alg_bytes);
/* doubling rounds summed over blocks and block count of the last compress call */
int bz2hip_last_stats(bz2hip_ctx *ctx, uint64_t *blocks, uint64_t *rounds_sum,
                      uint64_t *rle_bytes, uint64_t *periodic_blocks);

/* sum over blocks of n_b * rounds_b, and elements moved summed over all radix passes */
int bz2hip_last_stats2(bz2hip_ctx *ctx, uint64_t *n_times_rounds, uint64_t *sorted_elems);

/* host wall-clock seconds the last call spent driving the BWT stage (its round loop ends with a
 * device round trip, so this is the stage's elapsed time), and the number of periodic blocks on
 * which the reference's sorter would have indexed past its work array */
int bz2hip_last_stats3(bz2hip_ctx *ctx, double *bwt_seconds, uint64_t *tie_traps);

/* Periodic blocks since the context was created: replays of the reference's sorter actually run,
 * blocks that reused a result (same period word, k and declared block size: from the context's
 * cache or from a twin in the same batch), keys in the cache. */
int bz2hip_tie_stats(bz2hip_ctx *ctx, uint64_t *replays, uint64_t *reused, uint64_t *cached_keys);
/* Every origin pointer of a periodic block -- replayed, cached or shared with a twin -- is checked on
 * the device against the parallel sort (it must name one of the k rotations equal to rotation 0)
 * before it is written into a block header.  rejected = reused pointers that failed that check since
 * the context was created; their keys were dropped and the blocks replayed (0 unless two different
 * period words collided in the cache key). */
int bz2hip_tie_stats2(bz2hip_ctx *ctx, uint64_t *rejected);

#ifdef __cplusplus
}
#endif
#endif
