// ctx.hpp -- host-side context, workspace and launch helper of libbz2hip.
#pragma once
#include "common.hpp"
#include "../../include/bz2hip.h"

#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <tuple>
#include <vector>
#include <stdexcept>
#include <cstdio>

#define RCTL_SLOTS 4
#define H_CTL_WORDS 256     // page-locked control words (sort round loop, framing)
#define H_PIN_WORDS 32768   // page-locked scratch for small read-backs (block starts, bit offsets)

struct HipError : std::runtime_error {
    int code;
    HipError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            char _buf[512];                                                                  \
            snprintf(_buf, sizeof(_buf), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                     __FILE__, __LINE__);                                                    \
            throw HipError(_e == hipErrorOutOfMemory ? BZ2HIP_E_NOMEM : BZ2HIP_E_HIP, _buf); \
        }                                                                                    \
    } while (0)

// Carves arrays (256-byte aligned each) out of one device region; base == nullptr: count only.
struct Carver {
    u8 *base;        // nullptr: count only
    size_t off = 0;
    template <typename T>
    T *get(size_t count) {
        const size_t bytes = (std::max<size_t>(count * sizeof(T), 256) + 255) & ~(size_t)255;
        T *p = base ? (T *)(base + off) : nullptr;
        off += bytes;
        return p;
    }
};


// Periodic blocks (block == w^k): what identifies one for the origin-pointer cache, and a batch's
// pending decisions (k2_tie.hip).
struct TieKey {
    u32 level, n, k;
    u64 h1, h2;
    u64 w[8]; // the first 64 bytes of the period word, zero-padded: words up to 64 bytes compare exactly
    bool operator<(const TieKey &o) const {
        if (std::tie(level, n, k, h1, h2) != std::tie(o.level, o.n, o.k, o.h1, o.h2))
            return std::tie(level, n, k, h1, h2) < std::tie(o.level, o.n, o.k, o.h1, o.h2);
        for (int i = 0; i < 8; i++)
            if (w[i] != o.w[i]) return w[i] < o.w[i];
        return false;
    }
};
struct TieItem {
    int block;   // block of the batch
    int rep;     // block whose replay gives the value (== block: runs it itself), -1: from the cache
    u32 orig;
    TieKey key;
};

// Device workspace for one batch of up to NB blocks laid out at stride S elements.
struct Batch {
    int NB = 0;      // capacity (blocks)
    u32 S = 0;       // elements per block slot (multiple of RTILE), >= block_size + 8
    u32 TPB = 0;     // radix tiles per block slot = S / RTILE
    // block inputs
    u8 *T = nullptr;        // [NB][S]
    u32 *n = nullptr;       // [NB]
    u32 *present = nullptr; // [NB][8] bitmap words (bit v of word v>>5)
    u32 *crc = nullptr;     // [NB]
    // BWT
    u64 *keyA = nullptr, *keyB = nullptr; // round-0 / L-list radix ping-pong (element words E)
    u64 *slA = nullptr, *slB = nullptr;   // S-list ping-pong
    u32 *rank = nullptr;
    u32 *mS = nullptr, *mL = nullptr, *mSn = nullptr, *mLn = nullptr; // [NB] list lengths
    u32 *hcur = nullptr, *huse = nullptr, *state = nullptr, *par = nullptr, *leftg = nullptr;
    u32 *tilehist = nullptr;  // [NB][TPB][256]
    u32 *digitbase = nullptr; // [NB][256]
    u32 *tileagg = nullptr;   // [NB][TPB][4]
    u32 *pvote = nullptr;     // [NB][8192][2] distance votes {count, distance} (long-repeat rule)
    u32 *cand = nullptr;      // [NB][2] distances kept
    u32 *vagg = nullptr;      // [NB][2][TPB][2] per-tile min/max violation position
    u32 *runs = nullptr;      // [NB][68] regions {p, a, e, dir}
    u32 *nheads = nullptr;    // [NB]
    u32 *active = nullptr;    // [NB]
    u32 *rounds = nullptr;    // [NB]
    u32 *ctl = nullptr;       // [4] device control words (any_active, ...)
    u8 *bwt = nullptr;        // [NB][S]
    u32 *orig = nullptr;      // [NB]
    int level = 0;            // declared block size / 100000 the batch was sized for
    u32 *periodic = nullptr;  // [NB] period count k (1 = aperiodic)
    // MTF / RLE2
    u8 *mtfrec = nullptr;     // [NB][chunks][256] recency lists, then entry states
    u32 *mtfcnt = nullptr;    // [NB][chunks]
    u8 *mtfgrp = nullptr;     // [NB][32][256] group recency lists, then group entry states
    u32 *mtfglen = nullptr;   // [NB][32]
    u8 *mtfpos = nullptr;     // [NB][S]
    u32 *r2agg = nullptr;     // [NB][TPB][4] tile summaries (lz, tz, inner, allzero) -> offsets
    u16 *mtf = nullptr;       // [NB][S+?]  (slot size S2 = S + 64 halfwords)
    u32 *mtflen = nullptr;    // [NB]
    u32 *alpha = nullptr;     // [NB]
    u32 *freq = nullptr;      // [NB][258 -> 264]
    // Huffman
    u32 *ntables = nullptr;   // [NB]
    u8 *hlen = nullptr;       // [NB][6][264] code lengths
    u32 *hpart = nullptr;     // [NB][HS_PART_MAX][6][264] partial table frequencies
    u32 *hfreq = nullptr;     // [NB][6][264] final-pass table frequencies
    u32 *hcode = nullptr;     // [NB][6][264] (len<<24)|code
    u8 *sel = nullptr;        // [NB][SEL_STRIDE]
    u8 *selpos = nullptr;     // [NB][SEL_STRIDE] MTF'd selectors
    u32 *nsel = nullptr;      // [NB]
    u64 *tblbits = nullptr;   // [NB][6][2]: serialised table bits, data bits per table
    u64 *hdrbits = nullptr;   // [NB][4]: header bits, selector bits, table bits, data bits
    u64 *bitoff = nullptr;    // [2*(NB+1)] absolute bit offset of each block, then bit lengths
    u32 *pkagg = nullptr;     // [NB][S/PK_TILE] tile bit sums -> offsets
    size_t bytes = 0;
    std::vector<void *> allocs; // device allocations of this workspace
    std::vector<TieItem> tie_items; // periodic blocks of the batch in flight (host side)
    bool tie_running = false;
};

struct ProfRec {
    int cls;
    hipEvent_t e0, e1;
};

struct bz2hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bz2hip_opts opts{};
    std::string err;
    Batch batch;
    // second workspace + stream: compress_core alternates batches between the two "lanes" so that
    // MTF / Huffman of one batch run under the BWT of the next (swap_lane exchanges them)
    Batch batch_other;
    hipStream_t stream_other = nullptr;
    int active_lane = 0;
    void swap_lane() {
        std::swap(batch, batch_other);
        std::swap(stream, stream_other);
        active_lane ^= 1;
    }
    // window scratch for K1 (split / RLE1 / CRC)
    u32 *w_rs = nullptr, *w_P = nullptr, *w_agg1 = nullptr, *w_agg2 = nullptr;
    u32 *w_starts = nullptr;
    u32 *w_binfo = nullptr;  // per block: end of its first run, P just after it (k1_split -> k1_emit)
    u16 *w_P16 = nullptr;    // flushed-length prefix per input byte, tile-local (k1_rle1.hip)
    u32 *w_agg1r = nullptr, *w_agg2r = nullptr; // tile aggregates as produced (k1_scan_tiles reads these, writes w_agg1 / w_agg2)
    u32 *w_hint = nullptr;   // per 256-byte cell: bracket of the end of a block that starts there (k1_hints)
    u32 *w_cagg = nullptr;   // CRC tile partials [batch blocks][w_ETPB] (used on the other stream)
    void *w_region = nullptr;  // device region holding rs / P / aggregates / starts
    void *w_eregion = nullptr; // device region holding the emit / CRC tile partials
    // tuning knobs, read from the environment once in bz2hip_create (experiments only)
    int lanes = 2;           // BZ2HIP_LANES
    int debug = 0;           // BZ2HIP_DEBUG
    u32 plain_mask = 2;      // BZ2HIP_PLAIN
    u32 r0bits = 40;         // BZ2HIP_R0BITS
    u32 r0_sweeps = 2;       // BZ2HIP_RSWEEPS
    bool mtf_inv = true;
    bool r0_rank16 = true;   // round-0 re-rank with 16 consecutive slots per thread (experiments: BZ2HIP_R0RANK16=0)
    int split_hints = 1;     // k1_hints before the block chain: 0 never, 1 when more than 8 blocks lie ahead, 2 always (tests)     // BZ2HIP_MTF_INV: MTF replay as inversion counts (k3_mtf_replay_inv) instead of the loop over run heads
    bool radix_dig = true;   // BZ2HIP_RADIX_DIG: a scatter pass leaves the next pass's digits as a byte array
    bool r0_rule = true;     // BZ2HIP_R0RULE: the run rule already in the re-rank of round 0
    bool radix_stage = true;  // BZ2HIP_RADIX_STAGE: order the tile by digit in LDS before the stores
    u32 radix_bits = 8;      // BZ2HIP_RADIX_BITS: digit width of the LSD passes (8: 5 passes, 10: 4 passes)
    size_t stream_chunk = (size_t)64 << 20; // BZ2HIP_STREAM_CHUNK_MIB: streaming flush threshold per GPU
    size_t stream_least = 0;                // BZ2HIP_STREAM_LEAST_MIB: hand-over threshold while the worker is idle (0: half a chunk)
    size_t stream_first = (size_t)32 << 20;  // BZ2HIP_STREAM_FIRST_MIB: hand-over threshold of a stream's first job
    bool stream_sync = false;               // BZ2HIP_STREAM_SYNC: compress inside bz2hip_stream_write (experiments)
    u64 dec_pass_bytes = 0xfffffff0ull;     // BZ2HIP_DEC_PASS_MIB (tests): decoded bytes written per pass (u32 offsets)
    // decoder workspace, kept between calls (one region, grown on demand)
    size_t dec_consumed = 0;   // bytes of the input the last successfully decoded stream occupied
    void *dec_region = nullptr;
    size_t dec_region_cap = 0;
    bool fixed_ws = false;   // allocation-free call in progress: workspace carved from the static scratch
    u8 *scratch = nullptr;   // this device's address of the static scratch array (api.hip)
    hipEvent_t crc_done = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr; // S-sort || L-path inside a doubling round
    // round control of the sort (k2_bwt.hip): page-locked slots the control words of the last rounds
    // are copied to, and the events that say a slot is complete
    u32 *h_ctl = nullptr;
    u32 *h_pin = nullptr;   // H_PIN_WORDS words behind h_ctl (one allocation, made by bz2hip_create)
    hipEvent_t ev_ctl[RCTL_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t stream_tie = nullptr;  // replays of the reference's sorter for periodic blocks
    hipEvent_t ev_tie = nullptr;
    std::map<TieKey, u32> tie_cache;   // origin pointers of periodic blocks seen so far
    u64 st_tie_replays = 0, st_tie_cached = 0; // since the context was created
    u64 st_tie_rejected = 0;           // reused pointers that failed the in-class check and were replayed
    bool tie_poison = false;           // BZ2HIP_TIE_POISON (tests): store wrong pointers in the cache
    size_t w_cap = 0;        // window capacity in input bytes
    u32 w_maxblocks = 0;
    u32 w_ETPB = 0;          // emit tiles per block slot
    int w_eagg_nb = 0;
    // host -> device upload in flight (bz2hip_compress): run_scan issues it chunk by chunk on the
    // copy stream and scans every chunk as it lands
    struct Upload {
        bool active = false;
        const u8 *host = nullptr;
        u8 *dev = nullptr;
        size_t n = 0, chunk = 0;
    } up;
    hipStream_t stream_copy = nullptr;
    std::vector<hipEvent_t> up_events;
    hipEvent_t up_event(size_t k) {
        while (up_events.size() <= k) {
            hipEvent_t e;
            HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            up_events.push_back(e);
        }
        return up_events[k];
    }
    // staging buffers for the host-pointer entry points
    u8 *h_in = nullptr;  size_t h_in_cap = 0;
    u8 *h_out = nullptr; size_t h_out_cap = 0;
    // profiling
    bool profile = false;
    int profile_level = 0;   // 1 = bandwidth kernels only (few events), 2 = every kernel class
    std::vector<ProfRec> prof_pending;
    std::vector<hipEvent_t> ev_pool;
    double prof_ms[P_COUNT] = {0};
    u64 prof_launches[P_COUNT] = {0};
    double prof_bytes[P_COUNT] = {0};
    // stats of the last call
    u64 st_blocks = 0, st_rounds = 0, st_rle_bytes = 0, st_periodic = 0;
    double st_bwt_seconds = 0;
    u64 st_tie_traps = 0; // periodic blocks on which the reference would have indexed past its array
    u64 st_s_elems = 0;      // elements sorted by the LDS (S) path, summed over rounds
    u64 st_sorted_elems = 0; // sum over radix passes of elements moved
    u64 st_nr = 0;           // sum over blocks of n_b * doubling rounds_b
    // several GPUs behind one context: peers[k] drives opts.devices[k + 1] (this context is device 0
    // of the list); every device shifts its segment to its bit offset in d_shift before the download
    std::vector<bz2hip_ctx *> peers;
    u8 *d_shift = nullptr;
    size_t d_shift_cap = 0;
    // sharding state (bz2hip_shard_scan / _chain / _compress)
    const u8 *sh_in = nullptr;
    size_t sh_avail = 0;
    int sh_level = 0, sh_state = 0;
    bool sh_two = false;
    u32 sh_usable = 0;
    std::vector<u32> sh_starts;
    // streaming state
    int s_level = 0;
    bool s_open = false, s_finished = false;
    std::vector<u8> s_in;      // buffered, not yet compressed input (from s_in_pos on)
    size_t s_in_pos = 0;
    std::vector<u8> s_out;     // compressed bytes not yet drained
    size_t s_out_pos = 0;
    u32 s_stream_crc = 0;
    u32 s_bitbuf = 0;          // pending partial byte (left aligned in low 8 bits)
    int s_bitcnt = 0;
    // Streaming on a single-device context is asynchronous: bz2hip_stream_write fills one of two
    // page-locked chunk buffers and hands full ones to a worker thread, which uploads, compresses and
    // appends to s_out while the caller fills the other buffer (the reference's write() also returns
    // before a block is closed, S/BZip2OutputStream.swift:85-104).  sw.mu guards the job queue, s_out /
    // s_out_pos and the error slot.  Between bz2hip_stream_begin and the end of bz2hip_stream_finish
    // the device side of the context belongs to the worker.
    struct StreamWorker {
        struct Job { int buf; size_t n; bool final; };
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::deque<Job> q;
        bool busy[2] = {false, false}; // chunk buffer handed over and not yet consumed
        bool running = false, quit = false;
        int active = 0;                // jobs queued or in progress
        int err = 0;                   // first error of the stream (reported by the next call)
        std::string errmsg;
        u8 *pin[2] = {nullptr, nullptr};
        size_t pin_cap = 0;
        int cur = 0;                   // caller: the buffer being filled, and its fill
        size_t fill = 0;
        // worker: two page-locked landing buffers for the jobs' output.  A job's bytes are read by
        // bz2hip_stream_read straight from its landing buffer (oq lists them in order); only when both
        // buffers still hold unread bytes is the older one moved to the heap queue s_out.
        u8 *pin_out[2] = {nullptr, nullptr};
        size_t pin_out_cap = 0;
        bool out_busy[2] = {false, false};
        struct OutSeg { int buf; size_t off, len; };
        std::deque<OutSeg> oq;         // after s_out in stream order
        size_t oq_bytes = 0;
        // device input buffers, one per page-locked chunk buffer: the caller uploads every piece it
        // has copied into pin[k] to d_buf[k] + gap + offset at once (copy stream), so a job's input is
        // in HBM when the job starts; the bytes of an incomplete last block are carried into the gap
        // in front of the other buffer's data by the job that leaves them
        u8 *d_buf[2] = {nullptr, nullptr};
        size_t d_buf_cap = 0;          // bytes behind the gap
        size_t gap = 0;                // >= the input an incomplete block can hold
        hipEvent_t ev_up[2] = {nullptr, nullptr};
        size_t carry = 0;              // worker: carried bytes in front of the next job's buffer
        bool first_job = true;         // caller: nothing handed over yet in this stream
    } sw;
    bool s_async = false;

    hipEvent_t get_event() {
        if (!ev_pool.empty()) {
            hipEvent_t e = ev_pool.back();
            ev_pool.pop_back();
            return e;
        }
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        return e;
    }
    void prof_collect() {
        for (auto &r : prof_pending) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
                prof_ms[r.cls] += ms;
                prof_launches[r.cls] += 1;
            }
            ev_pool.push_back(r.e0);
            ev_pool.push_back(r.e1);
        }
        prof_pending.clear();
    }
};

// Launch helper: records a HIP event pair around the launch when profiling is on.
struct ProfScope {
    bz2hip_ctx *c;
    int cls;
    hipEvent_t e0 = nullptr;
    static bool heavy(int k) {
        return k == P_RADIX_HIST || k == P_RADIX_SCATTER || k == P_BWT_KEYS || k == P_BWT_SSORT ||
               k == P_MTF_REPLAY || k == P_BWT_RERANK || k == P_BWT_TIE;
    }
    ProfScope(bz2hip_ctx *ctx, int k, double alg_bytes = 0) : c(ctx), cls(k) {
        if (c->profile && (c->profile_level >= 2 || heavy(k))) {
            e0 = c->get_event();
            HIPCHK(hipEventRecord(e0, c->stream));
            c->prof_bytes[cls] += alg_bytes;
        }
    }
    ~ProfScope() {
        if (c->profile && e0) {
            hipEvent_t e1 = c->get_event();
            (void)hipEventRecord(e1, c->stream);
            c->prof_pending.push_back({cls, e0, e1});
        }
    }
};

#define KCHECK() HIPCHK(hipGetLastError())

// stage drivers (each in its own .hip file)
void run_bwt_n(bz2hip_ctx *c, int nb, u32 maxn);                       // k2_bwt.hip
void run_present(bz2hip_ctx *c, int nb, u32 maxn);                     // k2_bwt.hip
void run_tie_launch(bz2hip_ctx *c, int nb);                            // k2_tie.hip
void run_tie_finish(bz2hip_ctx *c, int nb);                            // k2_tie.hip
void run_mtf(bz2hip_ctx *c, int nb, u32 maxn);                         // k3_mtf.hip
void run_huffman(bz2hip_ctx *c, int nb, u32 maxmtf);                   // k4_huffman.hip
// writes blocks at absolute bit offsets into d_out (zeroed); first_bit = bit offset of block 0
void run_pack(bz2hip_ctx *c, int nb, u32 maxmtf, u8 *d_out, u64 first_bit, bool per_block,
              size_t bits_stride, size_t out_cap_bytes, std::vector<u64> *h_nbits); // k4_huffman.hip
// stats (optional, 4 words): sums over the batch of rounds, block lengths, length x rounds; periodic blocks
void run_frame(bz2hip_ctx *c, int nb, u8 *d_out, int level, bool header, bool trailer, u32 crc_in,
               u64 *end_bit, u32 *crc_out, u64 *stats = nullptr);                            // k4_huffman.hip
// K1 (k1_rle1.hip)
void k1_init_tables(bz2hip_ctx *c);
void run_scan(bz2hip_ctx *c, const u8 *d_in, u32 n, u32 *d_rs, u32 *d_P, u32 *d_agg1, u32 *d_agg2);
u32 run_chain(bz2hip_ctx *c, const u8 *d_in, u32 n, int level, const u32 *d_rs, const u32 *d_P, const u32 *d_agg2,
              u32 *d_starts, u32 max_blocks, u32 s0, u32 stop, std::vector<u32> &h_starts);
u32 run_split(bz2hip_ctx *c, const u8 *d_in, u32 n, int level, u32 *d_rs, u32 *d_P, u32 *d_agg1,
              u32 *d_agg2, u32 *d_starts, u32 max_blocks, std::vector<u32> &h_starts);
void run_rle1_emit(bz2hip_ctx *c, const u8 *d_in, const u32 *d_agg_head, const u32 *d_starts, u32 b0,
                   int nb, u32 max_in_len, u32 ETPB);
