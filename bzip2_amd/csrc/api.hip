// api.hip -- C ABI of libbz2hip (include/bz2hip.h): context, workspace, pipeline drivers.
#include "ctx.hpp"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>

// Larger batches are faster (768 MiB: 6.9 GB/s at 256 blocks per batch, 7.5 GB/s at 384+); a
// level-9 block costs ~37 MB of workspace, so 1024 blocks = 38 GB of the 288 GB.
#define DEFAULT_BATCH_BLOCKS 1024
#define WMAX (1u << 30) // window of input scanned at once (u32 prefix sums stay < 2^32)
#define STREAM_CHUNK ((size_t)64 << 20) // streaming: input buffered per GPU before a flush
#define UPLOAD_CHUNK ((size_t)32 << 20) // host-buffer entry points: upload granule scanned as it lands

// ------------------------------------------------------------------ workspace
// A batch workspace is ONE device region carved into its arrays (256-byte aligned each), so that
// it either exists completely or not at all: capacities are published only after the allocation
// has succeeded, and a failed allocation leaves an empty batch (NB = 0) that the next call
// rebuilds.  The same carving serves the allocation-free entry point (region = the static
// scratch array, bz2hip_compress_device_noalloc below).
static u32 slot_elems(int level) {
    return (u32)(((size_t)level * 100000 + 8 + RTILE - 1) / RTILE * RTILE);
}

// fills N's pointers from `base` (or only counts when base == nullptr); returns the bytes used
static size_t layout_batch(Batch &N, int level, int nb_blocks, u8 *base) {
    Carver A{base};
    const u32 S = slot_elems(level);
    N.NB = nb_blocks;
    N.S = S;
    N.level = level;
    N.TPB = S / RTILE;
    const size_t nb = (size_t)nb_blocks;
    const size_t S2 = S + 64;
    N.T = A.get<u8>(nb * S);
    N.n = A.get<u32>(nb);
    N.present = A.get<u32>(nb * 8);
    N.crc = A.get<u32>(nb);
    N.keyA = A.get<u64>(nb * S);
    N.keyB = A.get<u64>(nb * S);
    N.slA = A.get<u64>(nb * S);
    N.slB = A.get<u64>(nb * S);
    N.rank = A.get<u32>(nb * S);
    N.mS = A.get<u32>(nb); N.mL = A.get<u32>(nb); N.mSn = A.get<u32>(nb); N.mLn = A.get<u32>(nb);
    N.hcur = A.get<u32>(nb); N.huse = A.get<u32>(nb); N.state = A.get<u32>(nb);
    N.par = A.get<u32>(nb); N.leftg = A.get<u32>(nb);
    N.tilehist = A.get<u32>(nb * N.TPB * 1024); // up to 1024 bins per tile (10-bit digits)
    N.digitbase = A.get<u32>(nb * 1024);
    N.tileagg = A.get<u32>(nb * N.TPB * 4);
    N.pvote = A.get<u32>(nb * 8192 * 2);
    N.cand = A.get<u32>(nb * 2);
    N.vagg = A.get<u32>(nb * 2 * N.TPB * 2);
    N.runs = A.get<u32>(nb * 68);
    N.nheads = A.get<u32>(nb);
    N.active = A.get<u32>(nb);
    N.rounds = A.get<u32>(nb);
    N.ctl = A.get<u32>(64);
    N.bwt = A.get<u8>(nb * S);
    N.orig = A.get<u32>(nb);
    N.periodic = A.get<u32>(nb);
    N.mtfrec = A.get<u8>(nb * (S / MTF_CHUNK) * 256);
    N.mtfcnt = A.get<u32>(nb * (S / MTF_CHUNK));
    N.mtfgrp = A.get<u8>(nb * 32 * 256);
    N.mtfglen = A.get<u32>(nb * 32);
    N.mtfpos = A.get<u8>(nb * S);
    N.r2agg = A.get<u32>(nb * N.TPB * 4);
    N.mtf = A.get<u16>(nb * S2);
    N.mtflen = A.get<u32>(nb);
    N.alpha = A.get<u32>(nb);
    N.freq = A.get<u32>(nb * 264);
    N.ntables = A.get<u32>(nb);
    N.hlen = A.get<u8>(nb * MAX_TABLES * 264);
    N.hpart = A.get<u32>(nb * HS_PART_MAX * MAX_TABLES * 264);
    N.hfreq = A.get<u32>(nb * MAX_TABLES * 264);
    N.hcode = A.get<u32>(nb * MAX_TABLES * 264);
    N.sel = A.get<u8>(nb * SEL_STRIDE);
    N.selpos = A.get<u8>(nb * SEL_STRIDE);
    N.nsel = A.get<u32>(nb);
    N.tblbits = A.get<u64>(nb * MAX_TABLES * 2);
    N.hdrbits = A.get<u64>(nb * 4);
    N.bitoff = A.get<u64>(2 * (nb + 1));
    N.pkagg = A.get<u32>(nb * ((S2 + PK_TILE - 1) / PK_TILE));
    N.bytes = A.off;
    return A.off;
}

static void dfree_all(bz2hip_ctx *c) {
    for (void *p : c->batch.allocs) (void)hipFree(p);
    c->batch = Batch();
}

static void ensure_batch(bz2hip_ctx *c, int level, int nb_needed) {
    Batch &B = c->batch;
    const u32 S = slot_elems(level);
    if (c->fixed_ws) {
        // allocation-free call: the workspace was carved from the static scratch for this level
        if (B.NB < 1 || B.S != S) throw HipError(BZ2HIP_E_STATE, "internal: fixed workspace does not fit the level");
        return;
    }
    int cap = c->opts.max_batch_blocks > 0 ? c->opts.max_batch_blocks : DEFAULT_BATCH_BLOCKS;
    int want = std::min(cap, std::max(nb_needed, 1));
    if (B.NB >= want && B.S == S) return;
    c->sh_state = 0; // the workspace a shard scan sized is about to be replaced
    want = std::max(want, B.S == S ? B.NB : 0);
    // free the old batch (window scratch is tracked separately); the context holds an empty batch
    // (NB = 0) until the new region exists, so a failed allocation cannot leave stale capacities
    dfree_all(c);
    Batch N;
    const size_t bytes = layout_batch(N, level, want, nullptr);
    void *region = nullptr;
    HIPCHK(hipMalloc(&region, bytes)); // throws BZ2HIP_E_NOMEM; c->batch stays empty
    layout_batch(N, level, want, (u8 *)region);
    N.allocs.push_back(region);
    c->batch = std::move(N);
    HIPCHK(hipMemsetAsync(c->batch.ctl, 0, 64 * sizeof(u32), c->stream));
}

static void wfree(bz2hip_ctx *c) {
    c->w_cap = 0;
    c->w_maxblocks = 0;
    c->w_eagg_nb = 0;
    if (c->w_region) (void)hipFree(c->w_region);
    if (c->w_eregion) (void)hipFree(c->w_eregion);
    c->w_region = c->w_eregion = nullptr;
    c->w_rs = c->w_P = c->w_agg1 = c->w_agg2 = c->w_starts = c->w_cagg = nullptr;
    c->w_P16 = nullptr;
    c->w_binfo = nullptr;
    c->w_hint = nullptr;
    c->w_agg1r = c->w_agg2r = nullptr;
}

static u32 window_max_blocks(size_t wlen, int level) {
    return (u32)(wlen / ((size_t)level * 50000) + 4);
}

static u32 emit_tiles_per_block(int level) {
    return (u32)(((u64)level * 100000u * 52 + 2048) / 4096 + 2);
}

// window scratch: the cell records (last head before / flushed-length prefix at every 256th input
// byte; w_rs = Hc, w_P = Pc of k1_rle1.hip), the 16-bit tile-local prefix per input byte, tile
// aggregates, block starts
static size_t layout_window(bz2hip_ctx *c, size_t wlen, u32 mb, u8 *base) {
    Carver A{base};
    const size_t tiles = (wlen + 4095) / 4096 + 2;
    const size_t cells = wlen / 256 + 4;
    u32 *rs = A.get<u32>(cells), *P = A.get<u32>(cells);
    u16 *p16 = A.get<u16>(wlen + 16);
    u32 *a1 = A.get<u32>(tiles), *a2 = A.get<u32>(tiles), *st = A.get<u32>((size_t)mb + 8);
    u32 *bi = A.get<u32>(2 * ((size_t)mb + 8));
    u32 *hn = A.get<u32>(2 * cells);
    u32 *a1r = A.get<u32>(tiles), *a2r = A.get<u32>(tiles);
    if (base) { c->w_agg1r = a1r; c->w_agg2r = a2r; c->w_rs = rs; c->w_P = P; c->w_P16 = p16; c->w_binfo = bi; c->w_hint = hn; c->w_agg1 = a1; c->w_agg2 = a2; c->w_starts = st; }
    return A.off;
}
// per-batch CRC tile partials (the emit needs none since its tile offsets come from the scan)
static size_t layout_emit(bz2hip_ctx *c, int nb, u32 ETPB, u8 *base) {
    Carver A{base};
    u32 *g = A.get<u32>((size_t)nb * ETPB);
    if (base) c->w_cagg = g;
    return A.off;
}

static void ensure_window(bz2hip_ctx *c, size_t wlen, int level) {
    const u32 ETPB = emit_tiles_per_block(level);
    const u32 mb = window_max_blocks(wlen, level);
    if (c->fixed_ws) {
        if (c->w_cap < wlen || c->w_maxblocks < mb || c->w_eagg_nb < c->batch.NB || c->w_ETPB != ETPB)
            throw HipError(BZ2HIP_E_STATE, "internal: fixed workspace does not fit the window");
        return;
    }
    if (c->w_cap < wlen || c->w_maxblocks < mb) {
        c->w_cap = 0; // nothing is published until the new region exists
        c->w_maxblocks = 0;
        if (c->w_region) (void)hipFree(c->w_region);
        c->w_region = nullptr;
        HIPCHK(hipMalloc(&c->w_region, layout_window(c, wlen, mb, nullptr)));
        layout_window(c, wlen, mb, (u8 *)c->w_region);
        c->w_cap = wlen;
        c->w_maxblocks = mb;
    }
    if (c->w_eagg_nb < c->batch.NB || c->w_ETPB != ETPB) {
        c->w_eagg_nb = 0;
        if (c->w_eregion) (void)hipFree(c->w_eregion);
        c->w_eregion = nullptr;
        HIPCHK(hipMalloc(&c->w_eregion, layout_emit(c, c->batch.NB, ETPB, nullptr)));
        layout_emit(c, c->batch.NB, ETPB, (u8 *)c->w_eregion);
        c->w_eagg_nb = c->batch.NB;
        c->w_ETPB = ETPB;
    }
}

// (returns whether the input staging buffer was replaced: what had been uploaded into it is gone,
// even when the new buffer has the old one's address)
static bool ensure_staging(bz2hip_ctx *c, size_t in_bytes, size_t out_bytes) {
    bool in_replaced = false;
    if (c->h_in_cap < in_bytes) {
        in_replaced = true;
        c->h_in_cap = 0;
        if (c->h_in) (void)hipFree(c->h_in);
        c->h_in = nullptr;
        HIPCHK(hipMalloc((void **)&c->h_in, in_bytes + 64));
        c->h_in_cap = in_bytes;
    }
    if (c->h_out_cap < out_bytes) {
        c->h_out_cap = 0;
        if (c->h_out) (void)hipFree(c->h_out);
        c->h_out = nullptr;
        HIPCHK(hipMalloc((void **)&c->h_out, out_bytes + 64));
        c->h_out_cap = out_bytes;
    }
    return in_replaced;
}

// ------------------------------------------------------------------ error plumbing
static void stream_drain(bz2hip_ctx *c) {
    if (!c || !c->sw.running) return;
    std::unique_lock<std::mutex> lk(c->sw.mu);
    c->sw.cv.wait(lk, [&] { return c->sw.active == 0; });
}

template <typename F>
static int guarded(bz2hip_ctx *c, F f) {
    if (!c) return BZ2HIP_E_PARAM;
    stream_drain(c); // the device side of the context is the stream worker's while it has work
    try {
        HIPCHK(hipSetDevice(c->device));
        return f();
    } catch (const HipError &e) {
        c->err = e.what();
        (void)hipGetLastError();
        return e.code;
    } catch (const std::bad_alloc &) {
        c->err = "host allocation failed";
        return BZ2HIP_E_NOMEM;
    } catch (const std::exception &e) {
        c->err = e.what();
        return BZ2HIP_E_HIP;
    }
}

static int fail(bz2hip_ctx *c, int code, const char *msg) {
    c->err = msg;
    return code;
}

// ------------------------------------------------------------------ life cycle
extern "C" const char *bz2hip_version(void) { return "bz2hip 0.1 (gfx950)"; }

extern "C" size_t bz2hip_compress_bound(size_t n) {
    return n + n / 4 + 4096 + (n / 100000 + 1) * 2600;
}

static std::string g_create_err = "no error";

// Static scratch of the allocation-free entry point (bz2hip_compress_device_noalloc): every GPU
// the library runs on holds its own copy of this array, at its own address, until the process
// ends.  In host code the name is only a stand-in for hipGetSymbolAddress -- the kernels get the
// per-device address kept in bz2hip_ctx::scratch, never the name.
#define NOALLOC_SCRATCH_BYTES ((size_t)BZ2HIP_NOALLOC_SCRATCH_MIB << 20)
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];

static int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

extern "C" int bz2hip_create(bz2hip_ctx **out, const bz2hip_opts *opts) {
    if (!out) return BZ2HIP_E_PARAM;
    *out = nullptr;
    int ndev = 0;
    hipError_t e0 = hipGetDeviceCount(&ndev);
    if (e0 != hipSuccess || ndev == 0) {
        g_create_err = std::string("hipGetDeviceCount: ") + hipGetErrorString(e0);
        return BZ2HIP_E_HIP;
    }
    bz2hip_ctx *c = new (std::nothrow) bz2hip_ctx();
    if (!c) return BZ2HIP_E_NOMEM;
    if (opts) c->opts = *opts;
    else { c->opts = bz2hip_opts{}; c->opts.device = -1; }
    if (c->opts.n_devices < 0 || c->opts.n_devices > BZ2HIP_MAX_DEVICES) {
        g_create_err = "n_devices outside 0..8";
        delete c;
        return BZ2HIP_E_PARAM;
    }
    int dev = c->opts.n_devices >= 1 ? c->opts.devices[0] : c->opts.device;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    }
    if (dev >= ndev) { g_create_err = "no such device"; delete c; return BZ2HIP_E_PARAM; }
    c->device = dev;
    c->profile = c->opts.profile != 0;
    c->profile_level = c->opts.profile;
    const int side_prio = env_int("BZ2HIP_SIDE_PRIO", 0);
    if (hipSetDevice(dev) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess ||
        (side_prio ? hipStreamCreateWithPriority(&c->stream_other, hipStreamDefault, side_prio)
                   : hipStreamCreate(&c->stream_other)) != hipSuccess) {
        g_create_err = "hipSetDevice/hipStreamCreate failed";
        delete c;
        return BZ2HIP_E_HIP;
    }
    // the environment is read here, once per context (experiments only; never on the call path)
    c->lanes = env_int("BZ2HIP_LANES", 2);
    c->debug = env_int("BZ2HIP_DEBUG", 0);
    c->plain_mask = (u32)env_int("BZ2HIP_PLAIN", 2);
    c->r0bits = (u32)env_int("BZ2HIP_R0BITS", 40);
    c->r0_sweeps = (u32)std::min(16, std::max(1, env_int("BZ2HIP_RSWEEPS", 2)));
    c->radix_bits = env_int("BZ2HIP_RADIX_BITS", 8) == 10 ? 10u : 8u;
    c->radix_stage = env_int("BZ2HIP_RADIX_STAGE", 1) != 0;
    c->r0_rule = env_int("BZ2HIP_R0RULE", 1) != 0;
    c->radix_dig = env_int("BZ2HIP_RADIX_DIG", 1) != 0;
    c->mtf_inv = env_int("BZ2HIP_MTF_INV", 1) != 0;
    c->split_hints = env_int("BZ2HIP_SPLIT_HINTS", 1);
    c->r0_rank16 = env_int("BZ2HIP_R0RANK16", 1) != 0;
    c->tie_poison = env_int("BZ2HIP_TIE_POISON", 0) != 0; // tests: cached origin pointers are stored wrong
    c->stream_chunk = (size_t)std::max(1, env_int("BZ2HIP_STREAM_CHUNK_MIB", (int)(STREAM_CHUNK >> 20))) << 20;
    c->stream_sync = env_int("BZ2HIP_STREAM_SYNC", 0) != 0;
    c->stream_least = (size_t)std::max(0, env_int("BZ2HIP_STREAM_LEAST_MIB", 0)) << 20;
    c->stream_first = (size_t)std::max(1, env_int("BZ2HIP_STREAM_FIRST_MIB", 32)) << 20;
    if (int m = env_int("BZ2HIP_DEC_PASS_MIB", 0)) c->dec_pass_bytes = (u64)std::max(1, m) << 20;
    try {
        k1_init_tables(c);
        HIPCHK(hipEventCreateWithFlags(&c->crc_done, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->ev_tie, hipEventDisableTiming));
        HIPCHK(hipStreamCreate(&c->stream_tie));
        HIPCHK(hipStreamCreate(&c->stream_copy));
        for (int k = 0; k < RCTL_SLOTS; k++) HIPCHK(hipEventCreateWithFlags(&c->ev_ctl[k], hipEventDisableTiming));
        HIPCHK(hipHostMalloc((void **)&c->h_ctl, (H_CTL_WORDS + H_PIN_WORDS) * sizeof(u32), hipHostMallocDefault));
        c->h_pin = c->h_ctl + H_CTL_WORDS;
        // this device's copy of the static scratch (device `dev` is current)
        void *sp = nullptr;
        HIPCHK(hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch)));
        c->scratch = (u8 *)sp;
    } catch (const std::exception &e) {
        g_create_err = e.what();
        if (c->crc_done) (void)hipEventDestroy(c->crc_done);
        if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
        if (c->ev_join) (void)hipEventDestroy(c->ev_join);
        if (c->ev_tie) (void)hipEventDestroy(c->ev_tie);
        if (c->stream_tie) (void)hipStreamDestroy(c->stream_tie);
        if (c->stream_copy) (void)hipStreamDestroy(c->stream_copy);
        for (int k = 0; k < RCTL_SLOTS; k++)
            if (c->ev_ctl[k]) (void)hipEventDestroy(c->ev_ctl[k]);
        if (c->h_ctl) (void)hipHostFree(c->h_ctl);
        (void)hipStreamDestroy(c->stream);
        (void)hipStreamDestroy(c->stream_other);
        delete c;
        return BZ2HIP_E_HIP;
    }
    // one more single-device context per further entry of the device list
    for (int k = 1; k < c->opts.n_devices; k++) {
        bz2hip_opts po = c->opts;
        po.n_devices = 0;
        po.device = c->opts.devices[k] < 0 ? dev : c->opts.devices[k];
        bz2hip_ctx *peer = nullptr;
        int rc = bz2hip_create(&peer, &po);
        if (rc != BZ2HIP_OK) {
            bz2hip_destroy(c);
            return rc;
        }
        c->peers.push_back(peer);
    }
    (void)hipSetDevice(dev);
    *out = c;
    return BZ2HIP_OK;
}

extern "C" void bz2hip_destroy(bz2hip_ctx *c) {
    if (!c) return;
    if (c->sw.running) {
        {
            std::unique_lock<std::mutex> lk(c->sw.mu);
            c->sw.cv.wait(lk, [&] { return c->sw.active == 0; });
            c->sw.quit = true;
        }
        c->sw.cv.notify_all();
        c->sw.th.join();
        c->sw.running = false;
    }
    for (bz2hip_ctx *p : c->peers) bz2hip_destroy(p);
    c->peers.clear();
    (void)hipSetDevice(c->device);
    for (int k = 0; k < 2; k++)
        if (c->sw.pin[k]) (void)hipHostFree(c->sw.pin[k]);
    for (int k = 0; k < 2; k++) {
        if (c->sw.d_buf[k]) (void)hipFree(c->sw.d_buf[k]);
        if (c->sw.pin_out[k]) (void)hipHostFree(c->sw.pin_out[k]);
        if (c->sw.ev_up[k]) (void)hipEventDestroy(c->sw.ev_up[k]);
    }
    if (c->d_shift) (void)hipFree(c->d_shift);
    if (c->dec_region) (void)hipFree(c->dec_region);
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->stream_other);
    dfree_all(c);
    c->swap_lane();
    dfree_all(c);
    wfree(c);
    if (c->h_in) (void)hipFree(c->h_in);
    if (c->h_out) (void)hipFree(c->h_out);
    for (auto &r : c->prof_pending) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->crc_done) (void)hipEventDestroy(c->crc_done);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_tie) (void)hipEventDestroy(c->ev_tie);
    if (c->stream_tie) { (void)hipStreamSynchronize(c->stream_tie); (void)hipStreamDestroy(c->stream_tie); }
    if (c->stream_copy) { (void)hipStreamSynchronize(c->stream_copy); (void)hipStreamDestroy(c->stream_copy); }
    for (auto e : c->up_events) (void)hipEventDestroy(e);
    for (int k = 0; k < RCTL_SLOTS; k++)
        if (c->ev_ctl[k]) (void)hipEventDestroy(c->ev_ctl[k]);
    if (c->h_ctl) (void)hipHostFree(c->h_ctl);
    (void)hipStreamDestroy(c->stream);
    (void)hipStreamDestroy(c->stream_other);
    delete c;
}

extern "C" const char *bz2hip_last_error(const bz2hip_ctx *c) {
    return c ? c->err.c_str() : g_create_err.c_str(); // NULL: error of the last failed create
}

// ------------------------------------------------------------------ the pipeline
static void reset_stats(bz2hip_ctx *c) {
    c->st_blocks = c->st_rounds = c->st_rle_bytes = c->st_periodic = 0;
    c->st_sorted_elems = c->st_nr = c->st_s_elems = 0;
    c->st_bwt_seconds = 0;
    c->st_tie_traps = 0;
}

static int batch_cap(const bz2hip_ctx *c) {
    return c->opts.max_batch_blocks > 0 ? c->opts.max_batch_blocks : DEFAULT_BATCH_BLOCKS;
}

// Sizes the workspace for a window of wlen bytes (batches of equal size, two lanes when the window
// needs several batches) and launches the K1 scan over it.  Returns whether two lanes are used.
static bool prepare_window(bz2hip_ctx *c, const u8 *d_win, size_t wlen, int level) {
    const u32 bs = (u32)level * 100000u;
    // Splitting one batch in two costs more than the overlap returns (measured: 221 blocks as
    // 2 x 111 is 10 % slower), so the second lane is used only when the window needs several
    // batches anyway.
    const int est = (int)std::min<size_t>(wlen / (bs - 4) + 1, 1u << 20);
    const int cap = batch_cap(c);
    const bool two = est > cap && c->lanes >= 2 && !c->fixed_ws;
    // equal batches: ceil(est / number of batches), not cap + remainder
    const int per = (est + (est + cap - 1) / cap - 1) / ((est + cap - 1) / cap);
    ensure_batch(c, level, per);
    if (two) {
        c->swap_lane();
        ensure_batch(c, level, per);
        c->swap_lane();
    }
    ensure_window(c, wlen, level);
    // the window scratch (run records, prefixes, aggregates, block starts) is overwritten now: a
    // scan kept from bz2hip_shard_scan is gone (bz2hip_shard_chain / _compress then say E_STATE
    // instead of walking another input's records); bz2hip_shard_scan sets the state after this call
    c->sh_state = 0;
    run_scan(c, d_win, (u32)wlen, c->w_rs, c->w_P, c->w_agg1, c->w_agg2);
    return two;
}

struct FrameState {
    u8 *d_out;
    size_t out_cap;
    u64 *bitpos;
    u32 *stream_crc;
    int level;
    bool wrote_header; // false: the first frame writes "BZh" + level at bit 0
    bool any_frame;
    std::vector<u32> *crcs_out;
};

// Compresses blocks [0, usable) of a scanned window (starts = their window-relative offsets) into
// F.d_out at bit *F.bitpos: batches of equal size, alternating lanes; pack + frame of batch k (which
// need the bit cursor, i.e. host round trips) are issued after the BWT of batch k + 1 has been
// driven, so batch k's MTF / Huffman kernels overlap that BWT on the other stream.
static void compress_window_blocks(bz2hip_ctx *c, const u8 *d_win, const std::vector<u32> &starts, u32 usable,
                                   bool two, bool trailer_after_last, FrameState &F) {
    const u32 bs = (u32)F.level * 100000u;
    const u32 nbat = (usable + (u32)c->batch.NB - 1) / (u32)c->batch.NB;
    const u32 bsz = (usable + nbat - 1) / nbat;
    struct Pending { bool valid; int lane; int nb; bool is_last; } pend = {false, 0, 0, false};
    auto finish = [&](const Pending &P) {
        if (c->active_lane != P.lane) c->swap_lane();
        Batch &B = c->batch;
        const int nb = P.nb;
        run_pack(c, nb, bs + 1, F.d_out, *F.bitpos, false, 0, F.out_cap, nullptr);
        u64 endbit;
        u32 crc_out;
        u64 bst[4];
        run_frame(c, nb, F.d_out, F.level, !F.wrote_header, P.is_last, *F.stream_crc, &endbit, &crc_out, bst);
        F.wrote_header = true;
        F.any_frame = true;
        *F.bitpos = endbit;
        *F.stream_crc = crc_out;
        if (F.crcs_out) {
            size_t o = F.crcs_out->size();
            F.crcs_out->resize(o + nb);
            HIPCHK(hipMemcpy(F.crcs_out->data() + o, B.crc, nb * sizeof(u32), hipMemcpyDeviceToHost));
        }
        // stats: summed on the device by the framing kernel (one read-back instead of three arrays)
        c->st_blocks += (u64)nb;
        c->st_rounds += bst[0];
        c->st_rle_bytes += bst[1];
        c->st_nr += bst[2];
        c->st_periodic += bst[3];
    };
    const int lane0 = c->active_lane;
    for (u32 b0 = 0, k = 0; b0 < usable; b0 += bsz, k++) {
        const int lane = two ? (lane0 ^ (int)(k & 1)) : lane0;
        if (c->active_lane != lane) c->swap_lane();
        const int nb = (int)std::min<u32>(bsz, usable - b0);
        u32 max_in = 0;
        for (int b = 0; b < nb; b++) max_in = std::max(max_in, starts[b0 + b + 1] - starts[b0 + b]);
        run_rle1_emit(c, d_win, c->w_agg1, c->w_starts, b0, nb, max_in, c->w_ETPB);
        run_bwt_n(c, nb, bs);
        run_mtf(c, nb, bs);
        run_huffman(c, nb, bs + 1);
        const Pending cur = {true, lane, nb, trailer_after_last && (b0 + nb == usable)};
        if (two) {
            // the previous batch lives in the other lane: pack it now, keep this one pending
            if (pend.valid) finish(pend);
            pend = cur;
        } else {
            // one lane only (the block estimate can be short of the real count, so several
            // batches may still be needed): pack before the workspace is reused
            finish(cur);
        }
    }
    if (pend.valid) finish(pend);
    if (c->active_lane != lane0) c->swap_lane();
}

// Compresses blocks of d_in[0..n) into d_out (zeroed by the caller up to out_cap) starting at
// bit *bitpos.  Only complete blocks are taken unless `final`.  Returns bytes consumed.
static size_t compress_core(bz2hip_ctx *c, const u8 *d_in, size_t n, int level, u8 *d_out,
                            size_t out_cap, u64 *bitpos, u32 *stream_crc, bool header, bool final,
                            bool trailer = true, std::vector<u32> *crcs_out = nullptr) {
    size_t pos = 0;
    reset_stats(c);
    FrameState F{d_out, out_cap, bitpos, stream_crc, level, !header, false, crcs_out};
    std::vector<u32> starts;
    while (pos < n) {
        const size_t wlen = std::min<size_t>(n - pos, WMAX);
        const bool last_window = (pos + wlen == n);
        const bool two = prepare_window(c, d_in + pos, wlen, level);
        u32 cnt = run_chain(c, d_in + pos, (u32)wlen, level, c->w_rs, c->w_P, c->w_agg2, c->w_starts,
                            c->w_maxblocks, 0, (u32)wlen, starts);
        // a block that ends at the window end is complete only if this is the final data
        u32 usable = cnt;
        if (!(last_window && final) && cnt > 0 && starts[cnt] == (u32)wlen) usable = cnt - 1;
        if (usable == 0) {
            if (last_window) break; // leave the tail for the next call
            throw HipError(BZ2HIP_E_HIP, "window holds no complete block");
        }
        compress_window_blocks(c, d_in + pos, starts, usable, two, final && trailer && last_window, F);
        pos += starts[usable];
        if (usable < cnt && last_window) break;
    }
    if (final && trailer && !F.any_frame) {
        // empty input (or nothing left): header and/or trailer only
        ensure_batch(c, level, 1);
        u64 bp = *bitpos;
        HIPCHK(hipMemcpyAsync(c->batch.bitoff, &bp, sizeof(u64), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        u64 endbit;
        u32 crc_out;
        run_frame(c, 0, d_out, level, !F.wrote_header, true, *stream_crc, &endbit, &crc_out);
        *bitpos = endbit;
        *stream_crc = crc_out;
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipStreamSynchronize(c->stream_other));
    if (c->profile) c->prof_collect();
    if (c->debug)
        fprintf(stderr, "[bz2hip] blocks=%llu rle_bytes=%llu rounds_sum=%llu S-elems=%llu radix-elem-passes=%llu periodic=%llu\n",
                (unsigned long long)c->st_blocks, (unsigned long long)c->st_rle_bytes,
                (unsigned long long)c->st_rounds, (unsigned long long)c->st_s_elems,
                (unsigned long long)c->st_sorted_elems, (unsigned long long)c->st_periodic);
    return pos;
}

extern "C" int bz2hip_compress_device(bz2hip_ctx *c, const void *d_in, size_t n, int level,
                                      void *d_out, size_t cap, size_t *out_len) {
    return guarded(c, [&]() -> int {
        if (level < 1 || level > 9) return fail(c, BZ2HIP_E_PARAM, "Invalid BZip2 block size");
        if (!d_out || !out_len || (n && !d_in)) return fail(c, BZ2HIP_E_PARAM, "null pointer");
        if (((uintptr_t)d_out & 3) != 0) return fail(c, BZ2HIP_E_PARAM, "d_out must be 4-byte aligned");
        if (cap < 64) return fail(c, BZ2HIP_E_CAP, "output buffer too small");
        size_t zlen = std::min(cap, bz2hip_compress_bound(n)) & ~(size_t)3;
        HIPCHK(hipMemsetAsync(d_out, 0, zlen, c->stream));
        u64 bitpos = 32;
        u32 scrc = 0;
        compress_core(c, (const u8 *)d_in, n, level, (u8 *)d_out, zlen, &bitpos, &scrc, true, true);
        *out_len = (size_t)((bitpos + 7) / 8);
        return BZ2HIP_OK;
    });
}

// ------------------------------------------------------------------ allocation-free variant
// bz2hip_compress_device for callers that cannot allocate device memory (tight serving loops,
// harnesses that forbid hipMalloc): the whole workspace (K1 window scratch + one batch) is carved
// from this GPU's copy of the static `scratch` array.  No hipMalloc / hipFree on this path; events
// and streams were created by bz2hip_create.  NOT capturable into a HIP graph: the path makes
// blocking host round trips (block chain read-back, round control words, bit cursor).  Every call on a GPU shares
// that GPU's array, so calls that overlap on one GPU (two contexts, two threads) race.
extern "C" int bz2hip_compress_device_noalloc(bz2hip_ctx *c, const void *d_in, size_t n, int level,
                                              void *d_out, size_t cap, size_t *out_len) {
    return guarded(c, [&]() -> int {
        if (level < 1 || level > 9) return fail(c, BZ2HIP_E_PARAM, "Invalid BZip2 block size");
        if (!d_out || !out_len || (n && !d_in)) return fail(c, BZ2HIP_E_PARAM, "null pointer");
        if (((uintptr_t)d_out & 3) != 0) return fail(c, BZ2HIP_E_PARAM, "d_out must be 4-byte aligned");
        if (n > (size_t)BZ2HIP_NOALLOC_MAX_INPUT)
            return fail(c, BZ2HIP_E_PARAM, "input larger than BZ2HIP_NOALLOC_MAX_INPUT (1 MiB): use bz2hip_compress_device");
        if (cap < 64) return fail(c, BZ2HIP_E_CAP, "output buffer too small");
        if (!c->scratch) return fail(c, BZ2HIP_E_STATE, "static scratch not resolved on this device");
        if (c->fixed_ws) return fail(c, BZ2HIP_E_STATE, "allocation-free call already in progress");
        // the context's own (allocated) workspace is set aside for the duration of the call
        struct Saved {
            bz2hip_ctx *c;
            Batch batch;
            u32 *w_rs, *w_P, *w_agg1, *w_agg2, *w_starts, *w_cagg;
            u16 *w_P16;
            u32 *w_binfo, *w_hint, *w_agg1r, *w_agg2r;
            size_t w_cap;
            u32 w_maxblocks, w_ETPB;
            int w_eagg_nb;
            bool profile;
            explicit Saved(bz2hip_ctx *cc)
                : c(cc), batch(std::move(cc->batch)), w_rs(cc->w_rs), w_P(cc->w_P), w_agg1(cc->w_agg1),
                  w_agg2(cc->w_agg2), w_starts(cc->w_starts), w_cagg(cc->w_cagg),
                  w_P16(cc->w_P16), w_binfo(cc->w_binfo), w_hint(cc->w_hint), w_agg1r(cc->w_agg1r), w_agg2r(cc->w_agg2r), w_cap(cc->w_cap), w_maxblocks(cc->w_maxblocks), w_ETPB(cc->w_ETPB),
                  w_eagg_nb(cc->w_eagg_nb), profile(cc->profile) {
                cc->batch = Batch();
                cc->fixed_ws = true;
                cc->profile = false; // the event pool may have to create events
            }
            ~Saved() {
                c->batch = std::move(batch);
                c->w_rs = w_rs; c->w_P = w_P; c->w_agg1 = w_agg1; c->w_agg2 = w_agg2;
                c->w_starts = w_starts; c->w_cagg = w_cagg;
                c->w_P16 = w_P16;
                c->w_binfo = w_binfo;
                c->w_hint = w_hint;
                c->w_agg1r = w_agg1r; c->w_agg2r = w_agg2r;
                c->w_cap = w_cap; c->w_maxblocks = w_maxblocks; c->w_ETPB = w_ETPB;
                c->w_eagg_nb = w_eagg_nb;
                c->profile = profile;
                c->fixed_ws = false;
            }
        } saved(c);
        // carve: window scratch for n bytes, then as many block slots as fit
        const size_t wlen = std::max<size_t>(n, 1);
        const u32 mb = window_max_blocks(wlen, level);
        const u32 ETPB = emit_tiles_per_block(level);
        size_t off = layout_window(c, wlen, mb, c->scratch);
        int nb = (int)std::min<size_t>(wlen / ((size_t)level * 100000 - 4) + 1, 64);
        Batch N;
        while (nb >= 1 && off + layout_batch(N, level, nb, nullptr) + layout_emit(c, nb, ETPB, nullptr) >
                              NOALLOC_SCRATCH_BYTES)
            nb--;
        if (nb < 1) return fail(c, BZ2HIP_E_NOMEM, "static scratch too small for one block of this level");
        off += layout_batch(N, level, nb, c->scratch + off);
        off += layout_emit(c, nb, ETPB, c->scratch + off);
        c->batch = std::move(N);
        c->w_cap = wlen;
        c->w_maxblocks = mb;
        c->w_ETPB = ETPB;
        c->w_eagg_nb = nb;
        // the array keeps the last call's contents: the control words must start at zero
        HIPCHK(hipMemsetAsync(c->batch.ctl, 0, 64 * sizeof(u32), c->stream));
        size_t zlen = std::min(cap, bz2hip_compress_bound(n)) & ~(size_t)3;
        HIPCHK(hipMemsetAsync(d_out, 0, zlen, c->stream));
        u64 bitpos = 32;
        u32 scrc = 0;
        compress_core(c, (const u8 *)d_in, n, level, (u8 *)d_out, zlen, &bitpos, &scrc, true, true);
        *out_len = (size_t)((bitpos + 7) / 8);
        return BZ2HIP_OK;
    });
}

// ------------------------------------------------------------------ sharding one stream
// The K1 scan of a shard does not depend on where its first block starts, so every GPU scans its
// slice at once; only the block chain is serial across shards (a few dependent loads per block),
// and what crosses from one shard to the next is a single offset.
extern "C" int bz2hip_shard_scan(bz2hip_ctx *c, const void *d_in, size_t avail, int level) {
    return guarded(c, [&]() -> int {
        if (level < 1 || level > 9) return fail(c, BZ2HIP_E_PARAM, "Invalid BZip2 block size");
        if (avail && !d_in) return fail(c, BZ2HIP_E_PARAM, "null pointer");
        if (avail > WMAX) return fail(c, BZ2HIP_E_PARAM, "shard larger than one window (1 GiB): cut it in sub-shards");
        reset_stats(c);
        c->sh_in = (const u8 *)d_in;
        c->sh_avail = avail;
        c->sh_level = level;
        c->sh_usable = 0;
        c->sh_starts.clear();
        c->sh_two = avail ? prepare_window(c, c->sh_in, avail, level) : false;
        c->sh_state = 1;
        return BZ2HIP_OK;
    });
}

extern "C" int bz2hip_shard_chain(bz2hip_ctx *c, uint64_t first, uint64_t limit, int final,
                                  uint32_t *n_blocks, uint64_t *next) {
    return guarded(c, [&]() -> int {
        if (!n_blocks || !next) return fail(c, BZ2HIP_E_PARAM, "null pointer");
        if (c->sh_state < 1) return fail(c, BZ2HIP_E_STATE, "bz2hip_shard_scan has not run");
        if (first > c->sh_avail) return fail(c, BZ2HIP_E_PARAM, "first block start beyond the shard's data");
        if (limit > c->sh_avail) limit = c->sh_avail;
        c->sh_usable = 0;
        c->sh_starts.assign(1, (u32)first);
        *n_blocks = 0;
        *next = first;
        c->sh_state = 2;
        if (first >= limit) return BZ2HIP_OK; // the previous shard's last block swallowed this one
        u32 cnt = run_chain(c, c->sh_in, (u32)c->sh_avail, c->sh_level, c->w_rs, c->w_P, c->w_agg2,
                            c->w_starts, c->w_maxblocks, (u32)first, (u32)limit, c->sh_starts);
        // a block that runs into the end of the data is complete only if the stream ends there
        u32 usable = cnt;
        if (!final && cnt > 0 && c->sh_starts[cnt] == (u32)c->sh_avail) usable = cnt - 1;
        c->sh_usable = usable;
        *n_blocks = usable;
        *next = c->sh_starts[usable];
        return BZ2HIP_OK;
    });
}

extern "C" int bz2hip_shard_compress(bz2hip_ctx *c, void *d_out, size_t cap, uint64_t *nbits,
                                     uint32_t *crcs, uint32_t max_blocks, uint32_t *n_blocks) {
    return guarded(c, [&]() -> int {
        if (!d_out || !nbits || !crcs || !n_blocks) return fail(c, BZ2HIP_E_PARAM, "null pointer");
        if (((uintptr_t)d_out & 3) != 0) return fail(c, BZ2HIP_E_PARAM, "d_out must be 4-byte aligned");
        if (c->sh_state < 2) return fail(c, BZ2HIP_E_STATE, "bz2hip_shard_chain has not run");
        if (c->sh_usable > max_blocks) return fail(c, BZ2HIP_E_CAP, "more blocks than max_blocks");
        if (cap < 64) return fail(c, BZ2HIP_E_CAP, "output buffer too small");
        const u32 usable = c->sh_usable;
        const size_t in_bytes = usable ? (size_t)(c->sh_starts[usable] - c->sh_starts[0]) : 0;
        size_t zlen = std::min(cap, bz2hip_compress_bound(in_bytes)) & ~(size_t)3;
        HIPCHK(hipMemsetAsync(d_out, 0, zlen, c->stream));
        u64 bitpos = 0;
        u32 scrc = 0;
        std::vector<u32> cr;
        FrameState F{(u8 *)d_out, zlen, &bitpos, &scrc, c->sh_level, true, false, &cr};
        if (usable) compress_window_blocks(c, c->sh_in, c->sh_starts, usable, c->sh_two, false, F);
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipStreamSynchronize(c->stream_other));
        if (c->profile) c->prof_collect();
        for (size_t i = 0; i < cr.size(); i++) crcs[i] = cr[i];
        *n_blocks = (uint32_t)cr.size();
        *nbits = bitpos;
        c->sh_state = 1; // the scan stays valid: another chain + compress may follow
        return BZ2HIP_OK;
    });
}

// ------------------------------------------------------------------ decoder (kd_decode.hip)
int run_decompress(bz2hip_ctx *c, const u8 *d_in, size_t nbytes, u8 *d_out, size_t cap, size_t *produced,
                   u32 *blocks_out, bool multi = false, u32 *streams_out = nullptr);

extern "C" int bz2hip_decompress_device(bz2hip_ctx *c, const void *d_in, size_t n, void *d_out,
                                        size_t cap, size_t *produced) {
    return guarded(c, [&]() -> int {
        if (!produced || (n && !d_in) || (cap && !d_out)) return fail(c, BZ2HIP_E_PARAM, "null pointer");
        u32 nblk = 0;
        int rc = run_decompress(c, (const u8 *)d_in, n, (u8 *)d_out, cap, produced, &nblk);
        c->st_blocks = nblk;
        return rc;
    });
}

extern "C" int bz2hip_decompress(bz2hip_ctx *c, const uint8_t *in, size_t n, uint8_t *out, size_t cap,
                                 size_t *produced) {
    return guarded(c, [&]() -> int {
        if (!produced || (n && !in) || (cap && !out)) return fail(c, BZ2HIP_E_PARAM, "null pointer");
        ensure_staging(c, n, cap);
        if (n) HIPCHK(hipMemcpyAsync(c->h_in, in, n, hipMemcpyHostToDevice, c->stream));
        u32 nblk = 0;
        int rc = run_decompress(c, c->h_in, n, c->h_out, cap, produced, &nblk);
        c->st_blocks = nblk;
        if (*produced) {
            HIPCHK(hipMemcpyAsync(out, c->h_out, *produced, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
        }
        return rc;
    });
}

extern "C" int bz2hip_decompress_streams_device(bz2hip_ctx *c, const void *d_in, size_t n, void *d_out,
                                                size_t cap, size_t *produced, uint32_t *n_streams) {
    return guarded(c, [&]() -> int {
        if (!produced || !n_streams || (n && !d_in) || (cap && !d_out)) return fail(c, BZ2HIP_E_PARAM, "null pointer");
        u32 nblk = 0;
        int rc = run_decompress(c, (const u8 *)d_in, n, (u8 *)d_out, cap, produced, &nblk, true, n_streams);
        c->st_blocks = nblk;
        return rc;
    });
}

extern "C" int bz2hip_decompress_streams(bz2hip_ctx *c, const uint8_t *in, size_t n, uint8_t *out, size_t cap,
                                         size_t *produced, uint32_t *n_streams) {
    return guarded(c, [&]() -> int {
        if (!produced || !n_streams || (n && !in) || (cap && !out)) return fail(c, BZ2HIP_E_PARAM, "null pointer");
        ensure_staging(c, n, cap);
        if (n) HIPCHK(hipMemcpyAsync(c->h_in, in, n, hipMemcpyHostToDevice, c->stream));
        u32 nblk = 0;
        int rc = run_decompress(c, c->h_in, n, c->h_out, cap, produced, &nblk, true, n_streams);
        c->st_blocks = nblk;
        if (*produced) {
            HIPCHK(hipMemcpyAsync(out, c->h_out, *produced, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
        }
        return rc;
    });
}

extern "C" int bz2hip_compress_blocks_device(bz2hip_ctx *c, const void *d_in, size_t n, int level,
                                             void *d_out, size_t cap, uint64_t *nbits,
                                             uint32_t *crcs, uint32_t max_blocks,
                                             uint32_t *n_blocks) {
    return guarded(c, [&]() -> int {
        if (level < 1 || level > 9) return fail(c, BZ2HIP_E_PARAM, "Invalid BZip2 block size");
        if (!d_out || !nbits || !crcs || !n_blocks || (n && !d_in)) return fail(c, BZ2HIP_E_PARAM, "null pointer");
        if (((uintptr_t)d_out & 3) != 0) return fail(c, BZ2HIP_E_PARAM, "d_out must be 4-byte aligned");
        if (cap < 64) return fail(c, BZ2HIP_E_CAP, "output buffer too small");
        size_t zlen = std::min(cap, bz2hip_compress_bound(n)) & ~(size_t)3;
        HIPCHK(hipMemsetAsync(d_out, 0, zlen, c->stream));
        u64 bitpos = 0;
        u32 scrc = 0;
        std::vector<u32> cr;
        compress_core(c, (const u8 *)d_in, n, level, (u8 *)d_out, zlen, &bitpos, &scrc, false, true, false, &cr);
        if (cr.size() > max_blocks) return fail(c, BZ2HIP_E_CAP, "more blocks than max_blocks");
        for (size_t i = 0; i < cr.size(); i++) crcs[i] = cr[i];
        *n_blocks = (uint32_t)cr.size();
        *nbits = bitpos;
        return BZ2HIP_OK;
    });
}

extern "C" int bz2hip_host_alloc(void **p, size_t bytes) {
    if (!p) return BZ2HIP_E_PARAM;
    *p = nullptr;
    hipError_t e = hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return e == hipErrorOutOfMemory ? BZ2HIP_E_NOMEM : BZ2HIP_E_HIP;
    }
    return BZ2HIP_OK;
}

extern "C" void bz2hip_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

// ------------------------------------------------------------------ several GPUs behind one context
static void append_bits(u8 *out, u64 *pos, const u8 *src, u64 nbits);
static void append_value(u8 *out, u64 *pos, int cnt, u32 v);
void run_place(bz2hip_ctx *c, const u8 *d_src, u64 nbits, u8 *d_out, u64 dst_bit); // k4_huffman.hip
// One stream sharded over the context's device list (bz2hip_opts.n_devices): a worker thread per
// GPU uploads its slice of the caller's buffer, scans it, takes its turn in the block chain (the
// only serial step: one offset travels from worker to worker), compresses its blocks, shifts the
// segment to its final bit offset on the device and downloads it straight into the caller's
// output.  The calling thread orders everything: bit offsets, the CRC fold, the words two
// segments share, header and trailer.
struct ShardJob {
    bz2hip_ctx *x = nullptr;
    size_t lo = 0, hi = 0;        // nominal byte range of the round
    size_t next = 0;              // where the following shard continues (absolute)
    u64 nbits = 0, off = 0;       // segment length, final bit offset
    std::vector<u32> crcs;
    u8 edge[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}; // first / last word of the shifted segment
    u64 edge_word[2] = {0, 0};
    int n_edges = 0;
    int rc = BZ2HIP_OK;
};

// Transfers between the callers' pageable buffers and a device, and the workspace (re)allocations
// next to them, are taken one worker at a time per GPU.  Seen on this stack (ROCm 7.2): with three workers
// uploading at once, one worker's hipMemcpyAsync of 7.95 MB from pageable memory delivered its last
// 613 280 bytes and left the 7 MiB before them untouched (the stream decoded to zeros there); twice in
// ~40 runs of the test suite, only in the first process on a fresh machine.  The copies are short
// next to the compression they feed; the kernels of the workers still overlap.
static std::mutex g_pageable_mu[BZ2HIP_MAX_DEVICES]; // per GPU: workers on different GPUs do not wait for each other
static std::mutex &pageable_mu(const bz2hip_ctx *x) { return g_pageable_mu[(unsigned)x->device % BZ2HIP_MAX_DEVICES]; }

struct ChainHop {
    std::mutex m;
    std::condition_variable cv;
    std::vector<long long> first; // -1: not known yet, -2: the shard before failed
    void put(size_t r, long long v) {
        {
            std::lock_guard<std::mutex> g(m);
            if (r < first.size() && first[r] == -1) first[r] = v;
        }
        cv.notify_all();
    }
    long long get(size_t r) {
        std::unique_lock<std::mutex> g(m);
        cv.wait(g, [&] { return first[r] != -1; });
        return first[r];
    }
};

// phase 1 of one worker: upload, scan, chain, compress (segment stays in x->h_out)
static void shard_phase1(ShardJob &J, size_t r, size_t nw, const u8 *in, size_t n, int level, bool final_stream,
                         ChainHop &hop) {
    bz2hip_ctx *x = J.x;
    J.rc = guarded(x, [&]() -> int {
        const size_t bs = (size_t)level * 100000;
        const size_t maxlook = bs * 52 + 65536;
        size_t look = bs + 65536;
        size_t uploaded = 0;
        long long first = -1;
        u32 nblocks = 0;
        for (;;) {
            const size_t end = std::min(n, J.hi + look);
            const size_t avail = end - J.lo;
            if (avail > WMAX) return fail(x, BZ2HIP_E_PARAM, "internal: shard larger than one window");
            int rc;
            {
                std::lock_guard<std::mutex> g(pageable_mu(x));
                if (ensure_staging(x, avail, bz2hip_compress_bound(avail))) uploaded = 0;
                if (avail > uploaded) {
                    HIPCHK(hipMemcpyAsync(x->h_in + uploaded, in + J.lo + uploaded, avail - uploaded,
                                          hipMemcpyHostToDevice, x->stream));
                    HIPCHK(hipStreamSynchronize(x->stream));
                }
                uploaded = avail;
                rc = bz2hip_shard_scan(x, x->h_in, avail, level); // (allocates; its kernels are only queued here)
            }
            if (rc != BZ2HIP_OK) return rc;
            if (first < 0) first = hop.get(r);
            if (first == -2) return fail(x, BZ2HIP_E_STATE, "the shard before this one failed");
            // no block starts in this range: the previous shard's last block swallowed it, or (data
            // not final) that block ran into the end of the data and the chain stops before it
            if ((size_t)first >= J.hi || (size_t)first < J.lo) {
                nblocks = 0;
                J.next = (size_t)first;
                break;
            }
            uint64_t nxt = 0;
            rc = bz2hip_shard_chain(x, (u64)first - J.lo, J.hi - J.lo, (final_stream && end == n) ? 1 : 0,
                                    &nblocks, &nxt);
            if (rc != BZ2HIP_OK) return rc;
            J.next = J.lo + (size_t)nxt;
            if (J.next >= J.hi || end == n) break;
            if (look >= maxlook) return fail(x, BZ2HIP_E_HIP, "block chain did not close within the maximum lookahead");
            look = std::min(look * 4, maxlook); // the last block ran into the end of the slice
        }
        hop.put(r + 1, (long long)J.next);
        J.nbits = 0;
        J.crcs.clear();
        if (nblocks) {
            J.crcs.resize(nblocks);
            u32 got = 0;
            int rc = bz2hip_shard_compress(x, x->h_out, x->h_out_cap, &J.nbits, J.crcs.data(), nblocks, &got);
            if (rc != BZ2HIP_OK) return rc;
            J.crcs.resize(got);
        }
        return BZ2HIP_OK;
    });
    if (J.rc != BZ2HIP_OK) hop.put(r + 1, -2);
    (void)nw;
}

// phase 2 of one worker: shift the segment to bit (off & 31) on the device, download its inner
// words straight to their place in `out`, hand the two edge words to the calling thread
static void shard_phase2(ShardJob &J, u8 *out) {
    bz2hip_ctx *x = J.x;
    J.n_edges = 0;
    if (!J.nbits) return;
    J.rc = guarded(x, [&]() -> int {
        std::lock_guard<std::mutex> g(pageable_mu(x));
        const u64 sh = J.off & 31;
        const size_t W = (size_t)((sh + J.nbits + 31) / 32);
        if (x->d_shift_cap < W * 4) {
            x->d_shift_cap = 0;
            if (x->d_shift) (void)hipFree(x->d_shift);
            x->d_shift = nullptr;
            HIPCHK(hipMalloc((void **)&x->d_shift, W * 4 + (W * 4) / 4 + 4096));
            x->d_shift_cap = W * 4 + (W * 4) / 4 + 4096;
        }
        HIPCHK(hipMemsetAsync(x->d_shift, 0, W * 4, x->stream));
        run_place(x, x->h_out, J.nbits, x->d_shift, sh);
        const u64 w0 = J.off >> 5;
        if (W > 2)
            HIPCHK(hipMemcpyAsync(out + (w0 + 1) * 4, x->d_shift + 4, (W - 2) * 4, hipMemcpyDeviceToHost, x->stream));
        HIPCHK(hipMemcpyAsync(J.edge[0], x->d_shift, 4, hipMemcpyDeviceToHost, x->stream));
        J.edge_word[0] = w0;
        J.n_edges = 1;
        if (W > 1) {
            HIPCHK(hipMemcpyAsync(J.edge[1], x->d_shift + (W - 1) * 4, 4, hipMemcpyDeviceToHost, x->stream));
            J.edge_word[1] = w0 + W - 1;
            J.n_edges = 2;
        }
        HIPCHK(hipStreamSynchronize(x->stream));
        return BZ2HIP_OK;
    });
}

// Compresses in[0..n) on all devices of the context into `out` starting at bit *bitpos (bits before it
// inside its byte are kept).  Complete blocks only unless `final`; trailer only when `final`.
// Returns the bytes consumed.
static size_t multi_compress(bz2hip_ctx *c, const u8 *in, size_t n, int level, bool final, u8 *out,
                             size_t cap, u64 *bitpos, u32 *stream_crc) {
    std::vector<bz2hip_ctx *> devs;
    devs.push_back(c);
    for (bz2hip_ctx *p : c->peers) devs.push_back(p);
    const size_t bs = (size_t)level * 100000;
    struct { u64 blocks = 0, rounds = 0, rle = 0, periodic = 0, nr = 0, sorted = 0, s = 0, traps = 0; double bwt = 0; } tot;
    size_t pos = 0;
    // a round gives every worker at most ROUND_MAX bytes (+ lookahead <= one 1 GiB window)
    const size_t ROUND_MAX = (size_t)768 << 20;
    while (pos < n) {
        const size_t left = n - pos;
        size_t nw = std::min(devs.size(), std::max<size_t>(1, left / (2 * bs)));
        const size_t span = std::min(left, nw * ROUND_MAX);
        const bool last_round = (span == left);
        std::vector<ShardJob> jobs(nw);
        ChainHop hop;
        hop.first.assign(nw + 1, -1);
        hop.first[0] = (long long)pos;
        for (size_t r = 0; r < nw; r++) {
            jobs[r].x = devs[r];
            jobs[r].lo = pos + (r == 0 ? 0 : (span * r / nw) / 4096 * 4096);
            jobs[r].hi = (r + 1 == nw) ? pos + span : pos + (span * (r + 1) / nw) / 4096 * 4096;
        }
        const size_t vis = last_round ? n : std::min(n, pos + span + bs * 52 + 65536); // bytes a worker may look at
        {
            std::vector<std::thread> th;
            for (size_t r = 1; r < nw; r++)
                th.emplace_back(shard_phase1, std::ref(jobs[r]), r, nw, in, vis, level, final && last_round,
                                std::ref(hop));
            shard_phase1(jobs[0], 0, nw, in, vis, level, final && last_round, hop);
            for (auto &t : th) t.join();
        }
        for (size_t r = 0; r < nw; r++)
            if (jobs[r].rc != BZ2HIP_OK) {
                if (jobs[r].x != c) c->err = jobs[r].x->err;
                throw HipError(jobs[r].rc, c->err);
            }
        // ordered part 1: bit offsets, capacity, CRC fold
        u64 p = *bitpos;
        for (size_t r = 0; r < nw; r++) {
            jobs[r].off = p;
            p += jobs[r].nbits;
        }
        if ((p + 80 + 7) / 8 + 4 > cap) throw HipError(BZ2HIP_E_CAP, "output buffer too small");
        {
            std::vector<std::thread> th;
            for (size_t r = 1; r < nw; r++) th.emplace_back(shard_phase2, std::ref(jobs[r]), out);
            shard_phase2(jobs[0], out);
            for (auto &t : th) t.join();
        }
        for (size_t r = 0; r < nw; r++)
            if (jobs[r].rc != BZ2HIP_OK) {
                if (jobs[r].x != c) c->err = jobs[r].x->err;
                throw HipError(jobs[r].rc, c->err);
            }
        // ordered part 2: the words that two segments (or a segment and what was there) share
        const u8 keep = (*bitpos & 7) ? (u8)(out[*bitpos >> 3] & (0xff00u >> (*bitpos & 7))) : 0;
        for (size_t r = 0; r < nw; r++)
            for (int e = 0; e < jobs[r].n_edges; e++)
                for (int k = 0; k < 4; k++) {
                    const u64 idx = jobs[r].edge_word[e] * 4 + k;
                    if (idx >= (*bitpos >> 3) && idx < cap) out[idx] = 0;
                }
        if (*bitpos & 7) out[*bitpos >> 3] = keep;
        for (size_t r = 0; r < nw; r++)
            for (int e = 0; e < jobs[r].n_edges; e++)
                for (int k = 0; k < 4; k++) {
                    const u64 idx = jobs[r].edge_word[e] * 4 + k;
                    if (idx < cap) out[idx] |= jobs[r].edge[e][k];
                }
        double round_bwt = 0;
        for (size_t r = 0; r < nw; r++) {
            for (u32 v : jobs[r].crcs) *stream_crc = ((*stream_crc << 1) | (*stream_crc >> 31)) ^ v;
            bz2hip_ctx *x = jobs[r].x;
            tot.blocks += x->st_blocks; tot.rounds += x->st_rounds; tot.rle += x->st_rle_bytes;
            tot.periodic += x->st_periodic; tot.nr += x->st_nr; tot.sorted += x->st_sorted_elems;
            tot.s += x->st_s_elems; tot.traps += x->st_tie_traps;
            round_bwt = std::max(round_bwt, x->st_bwt_seconds);
        }
        tot.bwt += round_bwt;
        *bitpos = p;
        const size_t consumed = jobs[nw - 1].next - pos;
        pos = jobs[nw - 1].next;
        if (consumed == 0) break; // nothing but an incomplete block left
        if (last_round) break;
    }
    c->st_blocks = tot.blocks; c->st_rounds = tot.rounds; c->st_rle_bytes = tot.rle; c->st_periodic = tot.periodic;
    c->st_nr = tot.nr; c->st_sorted_elems = tot.sorted; c->st_s_elems = tot.s; c->st_tie_traps = tot.traps;
    c->st_bwt_seconds = tot.bwt;
    if (final) {
        if ((*bitpos + 80 + 7) / 8 > cap) throw HipError(BZ2HIP_E_CAP, "output buffer too small");
        if ((*bitpos & 7) == 0) out[*bitpos >> 3] = 0;
        append_value(out, bitpos, 24, 0x177245);
        append_value(out, bitpos, 24, 0x385090);
        append_value(out, bitpos, 16, *stream_crc >> 16);
        append_value(out, bitpos, 16, *stream_crc & 0xffff);
    }
    return pos;
}

extern "C" int bz2hip_compress(bz2hip_ctx *c, const uint8_t *in, size_t n, int level, uint8_t *out,
                               size_t cap, size_t *out_len) {
    return guarded(c, [&]() -> int {
        if (level < 1 || level > 9) return fail(c, BZ2HIP_E_PARAM, "Invalid BZip2 block size");
        if (!out || !out_len || (n && !in)) return fail(c, BZ2HIP_E_PARAM, "null pointer");
        if (!c->peers.empty() && n >= (size_t)level * 400000) {
            // several GPUs: one stream sharded over the device list
            if (cap < 64) return fail(c, BZ2HIP_E_CAP, "output buffer too small");
            out[0] = 0x42; out[1] = 0x5a; out[2] = 0x68; out[3] = (u8)('0' + level);
            u64 bitpos = 32;
            u32 scrc = 0;
            multi_compress(c, in, n, level, true, out, cap, &bitpos, &scrc);
            *out_len = (size_t)((bitpos + 7) / 8);
            return BZ2HIP_OK;
        }
        size_t bound = bz2hip_compress_bound(n);
        ensure_staging(c, n, bound);
        if (n && n <= WMAX && n > 2 * UPLOAD_CHUNK) {
            // one window: the upload is issued chunk by chunk inside the K1 scan (run_scan), so that
            // the scan of a chunk runs under the transfer of the next
            c->up.active = true;
            c->up.host = in;
            c->up.dev = c->h_in;
            c->up.n = n;
            c->up.chunk = UPLOAD_CHUNK;
        } else if (n) {
            HIPCHK(hipMemcpyAsync(c->h_in, in, n, hipMemcpyHostToDevice, c->stream));
        }
        size_t got = 0;
        int rc = bz2hip_compress_device(c, c->h_in, n, level, c->h_out, bound, &got);
        c->up.active = false;
        if (rc != BZ2HIP_OK) return rc;
        if (got > cap) return fail(c, BZ2HIP_E_CAP, "output buffer too small");
        HIPCHK(hipMemcpy(out, c->h_out, got, hipMemcpyDeviceToHost));
        *out_len = got;
        return BZ2HIP_OK;
    });
}

extern "C" int bz2hip_compress_blocks(bz2hip_ctx *c, const uint8_t *in, size_t n, int level,
                                      uint8_t *out, size_t cap, uint64_t *nbits, uint32_t *crcs,
                                      uint32_t max_blocks, uint32_t *n_blocks) {
    return guarded(c, [&]() -> int {
        if (!out || !nbits || (n && !in)) return fail(c, BZ2HIP_E_PARAM, "null pointer");
        size_t bound = bz2hip_compress_bound(n);
        ensure_staging(c, n, bound);
        if (n) HIPCHK(hipMemcpyAsync(c->h_in, in, n, hipMemcpyHostToDevice, c->stream));
        int rc = bz2hip_compress_blocks_device(c, c->h_in, n, level, c->h_out, bound, nbits, crcs,
                                               max_blocks, n_blocks);
        if (rc != BZ2HIP_OK) return rc;
        size_t got = (size_t)((*nbits + 7) / 8);
        if (got > cap) return fail(c, BZ2HIP_E_CAP, "output buffer too small");
        if (got) HIPCHK(hipMemcpy(out, c->h_out, got, hipMemcpyDeviceToHost));
        return BZ2HIP_OK;
    });
}

// Appends nbits bits of src (MSB first) at bit *pos of out.  The bytes out[*pos/8 + 1 ...] are
// overwritten; the byte that holds bit *pos keeps its high (*pos & 7) bits.  64 bits per step.
static void append_bits(u8 *out, u64 *pos, const u8 *src, u64 nbits) {
    if (!nbits) return;
    u8 *p = out + (*pos >> 3);
    const int sh = (int)(*pos & 7);
    const u64 full = nbits >> 3;
    const int rem = (int)(nbits & 7);
    if (sh == 0) {
        memcpy(p, src, full);
        if (rem) p[full] = (u8)(src[full] & (0xff00u >> rem));
    } else {
        u64 carry = (u64)(p[0] >> (8 - sh)); // the sh pending bits, low-aligned
        u64 k = 0;
        for (; k + 8 <= full; k += 8) {
            u64 v;
            memcpy(&v, src + k, 8);
            v = __builtin_bswap64(v);
            u64 o = __builtin_bswap64((carry << (64 - sh)) | (v >> sh));
            memcpy(p + k, &o, 8);
            carry = v & ((1ull << sh) - 1ull);
        }
        // tail: bytes, then the last partial byte
        u32 acc = (u32)carry;
        int nb = sh;
        u8 *q = p + k;
        for (; k < full; k++) {
            acc = (acc << 8) | src[k];
            *q++ = (u8)(acc >> nb);
            acc &= (1u << nb) - 1u;
        }
        if (rem) {
            acc = (acc << rem) | (u32)(src[full] >> (8 - rem));
            nb += rem;
            if (nb >= 8) {
                *q++ = (u8)(acc >> (nb - 8));
                nb -= 8;
                acc &= (1u << nb) - 1u;
            }
        }
        if (nb > 0) *q = (u8)(acc << (8 - nb));
    }
    *pos += nbits;
}

static void append_value(u8 *out, u64 *pos, int cnt, u32 v) {
    const u32 w = v << (32 - cnt); // the cnt low bits of v, left-aligned
    const u8 be[4] = {(u8)(w >> 24), (u8)(w >> 16), (u8)(w >> 8), (u8)w};
    append_bits(out, pos, be, (u64)cnt);
}

// Host-side ordered concatenation of block segments into one stream (K8 across GPUs):
// S/BZip2OutputStream.swift:139 (CRC fold), :154-157 (trailer), :181-184 (header),
// S/BZip2BitOutputStream.swift:116-120 (zero pad).  Pure host code, no context needed.
extern "C" int bz2hip_stitch(int level, uint32_t nseg, const uint8_t *const *seg_bits,
                             const uint64_t *seg_nbits, const uint32_t *const *seg_crcs,
                             const uint32_t *seg_ncrc, uint8_t *out, size_t cap, size_t *out_len) {
    if (level < 1 || level > 9 || !out || !out_len) return BZ2HIP_E_PARAM;
    if (nseg && (!seg_bits || !seg_nbits || !seg_crcs || !seg_ncrc)) return BZ2HIP_E_PARAM;
    u64 total = 32 + 80;
    for (u32 i = 0; i < nseg; i++) total += seg_nbits[i];
    if ((total + 7) / 8 > cap) return BZ2HIP_E_CAP;
    u64 pos = 0;
    out[0] = 0;
    append_value(out, &pos, 16, 0x425a);
    append_value(out, &pos, 8, 0x68);
    append_value(out, &pos, 8, (u32)('0' + level));
    u32 scrc = 0;
    for (u32 i = 0; i < nseg; i++) {
        if ((pos & 7) == 0) out[pos >> 3] = 0;
        append_bits(out, &pos, seg_bits[i], seg_nbits[i]);
        for (u32 k = 0; k < seg_ncrc[i]; k++) scrc = ((scrc << 1) | (scrc >> 31)) ^ seg_crcs[i][k];
    }
    if ((pos & 7) == 0) out[pos >> 3] = 0;
    append_value(out, &pos, 24, 0x177245);
    append_value(out, &pos, 24, 0x385090);
    append_value(out, &pos, 16, scrc >> 16);
    append_value(out, &pos, 16, scrc & 0xffff);
    *out_len = (size_t)((pos + 7) >> 3);
    return BZ2HIP_OK;
}

// The same concatenation with the segments and the stream resident in HBM: every segment is
// funnel-shifted to its bit offset by a kernel (k8_place), then header and trailer are written.
void run_place(bz2hip_ctx *c, const u8 *d_src, u64 nbits, u8 *d_out, u64 dst_bit); // k4_huffman.hip
void run_ends(bz2hip_ctx *c, u8 *d_out, int level, u64 end_bit, u32 scrc);          // k4_huffman.hip

extern "C" int bz2hip_stitch_device(bz2hip_ctx *c, int level, uint32_t nseg,
                                    const void *const *d_seg_bits, const uint64_t *seg_nbits,
                                    const uint32_t *const *seg_crcs, const uint32_t *seg_ncrc,
                                    void *d_out, size_t cap, size_t *out_len) {
    return guarded(c, [&]() -> int {
        if (level < 1 || level > 9) return fail(c, BZ2HIP_E_PARAM, "Invalid BZip2 block size");
        if (!d_out || !out_len) return fail(c, BZ2HIP_E_PARAM, "null pointer");
        if (nseg && (!d_seg_bits || !seg_nbits || !seg_crcs || !seg_ncrc)) return fail(c, BZ2HIP_E_PARAM, "null pointer");
        if (((uintptr_t)d_out & 3) != 0) return fail(c, BZ2HIP_E_PARAM, "d_out must be 4-byte aligned");
        u64 total = 32 + 80;
        for (u32 i = 0; i < nseg; i++) {
            if (seg_nbits[i] && (((uintptr_t)d_seg_bits[i]) & 3) != 0)
                return fail(c, BZ2HIP_E_PARAM, "segments must be 4-byte aligned");
            total += seg_nbits[i];
        }
        const size_t words = (size_t)((total + 31) / 32);
        if (words * 4 > cap) return fail(c, BZ2HIP_E_CAP, "output buffer too small");
        HIPCHK(hipMemsetAsync(d_out, 0, words * 4, c->stream));
        u64 pos = 32;
        u32 scrc = 0;
        for (u32 i = 0; i < nseg; i++) {
            if (seg_nbits[i]) run_place(c, (const u8 *)d_seg_bits[i], seg_nbits[i], (u8 *)d_out, pos);
            pos += seg_nbits[i];
            for (u32 k = 0; k < seg_ncrc[i]; k++) scrc = ((scrc << 1) | (scrc >> 31)) ^ seg_crcs[i][k];
        }
        run_ends(c, (u8 *)d_out, level, pos, scrc);
        HIPCHK(hipStreamSynchronize(c->stream));
        *out_len = (size_t)((pos + 80 + 7) / 8);
        return BZ2HIP_OK;
    });
}

// ------------------------------------------------------------------ streaming
extern "C" int bz2hip_stream_begin(bz2hip_ctx *c, int level) {
    if (!c) return BZ2HIP_E_PARAM;
    if (level < 1 || level > 9) return fail(c, BZ2HIP_E_PARAM, "Invalid BZip2 block size");
    stream_drain(c); // what an abandoned stream still had in flight
    c->s_async = c->peers.empty() && !c->stream_sync;
    c->sw.cur = 0;
    c->sw.fill = 0;
    c->sw.carry = 0;
    c->sw.err = 0;
    c->sw.first_job = true;
    c->sw.oq.clear();
    c->sw.oq_bytes = 0;
    c->sw.out_busy[0] = c->sw.out_busy[1] = false;
    c->s_level = level;
    c->s_open = true;
    c->s_finished = false;
    c->s_in.clear();
    c->s_in_pos = 0;
    c->s_out.clear();
    c->s_out_pos = 0;
    c->s_stream_crc = 0;
    c->s_bitbuf = 0;
    c->s_bitcnt = 0;
    // the stream header reaches the sink at construction time (S/BZip2OutputStream.swift:181-184)
    const u8 hdr[4] = {0x42, 0x5a, 0x68, (u8)('0' + level)};
    c->s_out.insert(c->s_out.end(), hdr, hdr + 4);
    return BZ2HIP_OK;
}

// compress what is buffered; only complete blocks unless final
static int stream_flush(bz2hip_ctx *c, bool final) {
    return guarded(c, [&]() -> int {
        const u8 *src = c->s_in.data() + c->s_in_pos;
        const size_t n = c->s_in.size() - c->s_in_pos;
        size_t bound = bz2hip_compress_bound(n) + 16;
        u64 bitpos = (u64)c->s_bitcnt; // the stream header went out with bz2hip_stream_begin
        u32 scrc = c->s_stream_crc;
        std::vector<u8> tmp;
        size_t used;
        if (!c->peers.empty() && n >= (size_t)c->s_level * 400000) {
            tmp.resize(bound);
            tmp[0] = (u8)c->s_bitbuf;
            used = multi_compress(c, src, n, c->s_level, final, tmp.data(), tmp.size(), &bitpos, &scrc);
        } else {
            ensure_staging(c, n, bound);
            if (n) HIPCHK(hipMemcpyAsync(c->h_in, src, n, hipMemcpyHostToDevice, c->stream));
            size_t zlen = bound & ~(size_t)3;
            HIPCHK(hipMemsetAsync(c->h_out, 0, zlen, c->stream));
            used = compress_core(c, c->h_in, n, c->s_level, c->h_out, zlen, &bitpos, &scrc, false, final);
            if (final || used > 0) {
                tmp.resize((size_t)((bitpos + 7) / 8));
                if (!tmp.empty()) HIPCHK(hipMemcpy(tmp.data(), c->h_out, tmp.size(), hipMemcpyDeviceToHost));
                if (c->s_bitcnt > 0) tmp[0] |= (u8)c->s_bitbuf;
            }
        }
        c->s_stream_crc = scrc;
        if (final || used > 0) {
            const size_t nbytes = (size_t)((bitpos + 7) / 8);
            const size_t full = final ? nbytes : (size_t)(bitpos / 8);
            c->s_out.insert(c->s_out.end(), tmp.begin(), tmp.begin() + full);
            if (!final) {
                c->s_bitcnt = (int)(bitpos & 7);
                c->s_bitbuf = c->s_bitcnt ? tmp[full] : 0;
            } else {
                c->s_bitcnt = 0;
                c->s_bitbuf = 0;
            }
        }
        // consumed input is dropped lazily: no quadratic erase per flush
        c->s_in_pos += used;
        if (c->s_in_pos == c->s_in.size()) {
            c->s_in.clear();
            c->s_in_pos = 0;
        } else if (c->s_in_pos > (c->s_in.size() >> 1)) {
            c->s_in.erase(c->s_in.begin(), c->s_in.begin() + c->s_in_pos);
            c->s_in_pos = 0;
        }
        return BZ2HIP_OK;
    });
}

// ---- the asynchronous path (single-device contexts)
// One chunk on the worker thread: [carried bytes | the chunk] -> complete blocks (all of it when
// final) -> s_out.  The bytes of an incomplete last block stay in HBM, at the front of sw.d_in (a buffer of the
// stream's own: other calls on the context between two writes leave it alone).
static void stream_job(bz2hip_ctx *c, const bz2hip_ctx::StreamWorker::Job &job) {
    auto &W = c->sw;
    const auto t_job = std::chrono::steady_clock::now();
    const size_t n = W.carry + job.n;
    const size_t bound = bz2hip_compress_bound(n) + 16;
    if (W.carry > W.gap) throw HipError(BZ2HIP_E_HIP, "internal: carried bytes exceed the gap in front of the stream buffer");
    {
        // size the output staging and the batch workspace for the largest job this stream can hand
        // over, once, instead of growing them job by job (a few ms of hipFree / hipMalloc each time)
        const size_t nmax = std::max(n, W.pin_cap + (size_t)c->s_level * 200000u);
        ensure_staging(c, 0, std::max(bound, bz2hip_compress_bound(nmax) + 16));
        const u32 bs = (u32)c->s_level * 100000u;
        const int est = (int)std::min<size_t>(nmax / (bs - 4) + 1, (size_t)batch_cap(c));
        if (!c->fixed_ws) ensure_batch(c, c->s_level, est);
    }
    const auto t_a = std::chrono::steady_clock::now();
    // the job's input: [carried bytes | the chunk], both already in HBM (the caller uploaded the
    // chunk piece by piece while it filled it; the job before this one left the carried bytes)
    // (a stream that is finished before anything was written has no buffers yet: n == 0 then)
    const u8 *d_input = W.d_buf[job.buf] ? W.d_buf[job.buf] + W.gap - W.carry : nullptr;
    if (n && !d_input) throw HipError(BZ2HIP_E_HIP, "internal: stream job without a device buffer");
    if (W.ev_up[job.buf]) HIPCHK(hipStreamWaitEvent(c->stream, W.ev_up[job.buf], 0));
    const size_t zlen = bound & ~(size_t)3;
    HIPCHK(hipMemsetAsync(c->h_out, 0, zlen, c->stream));
    u64 bitpos = (u64)c->s_bitcnt;
    u32 scrc = c->s_stream_crc;
    const size_t used = compress_core(c, d_input, n, c->s_level, c->h_out, zlen, &bitpos, &scrc, false, job.final);
    c->s_stream_crc = scrc;
    const auto t_b = std::chrono::steady_clock::now();
    if (job.final || used > 0) {
        // download into a page-locked landing buffer kept by the stream; bz2hip_stream_read takes the
        // bytes from there (a copy into a heap queue cost as much again as the transfer)
        const size_t tlen = (size_t)((bitpos + 7) / 8);
        if (W.pin_out_cap < tlen + 8) {
            {
                // both buffers go: what they still hold moves to the heap queue first
                std::lock_guard<std::mutex> lk(W.mu);
                for (const auto &sg : W.oq) c->s_out.insert(c->s_out.end(), W.pin_out[sg.buf] + sg.off, W.pin_out[sg.buf] + sg.off + sg.len);
                W.oq.clear();
                W.oq_bytes = 0;
                W.out_busy[0] = W.out_busy[1] = false;
            }
            for (int k = 0; k < 2; k++) {
                if (W.pin_out[k]) (void)hipHostFree(W.pin_out[k]);
                W.pin_out[k] = nullptr;
            }
            W.pin_out_cap = 0;
            for (int k = 0; k < 2; k++) HIPCHK(hipHostMalloc((void **)&W.pin_out[k], c->h_out_cap + 64, hipHostMallocDefault));
            W.pin_out_cap = c->h_out_cap + 64;
        }
        int ob;
        {
            std::lock_guard<std::mutex> lk(W.mu);
            if (W.out_busy[0] && W.out_busy[1]) {
                // the reader is behind: the older buffer's unread bytes go to the heap queue
                const int old = W.oq.front().buf;
                while (!W.oq.empty() && W.oq.front().buf == old) {
                    const auto sg = W.oq.front();
                    c->s_out.insert(c->s_out.end(), W.pin_out[old] + sg.off, W.pin_out[old] + sg.off + sg.len);
                    W.oq_bytes -= sg.len;
                    W.oq.pop_front();
                }
                W.out_busy[old] = false;
            }
            ob = W.out_busy[0] ? 1 : 0;
            W.out_busy[ob] = true;
        }
        u8 *tmp = W.pin_out[ob];
        if (tlen) HIPCHK(hipMemcpy(tmp, c->h_out, tlen, hipMemcpyDeviceToHost));
        if (c->s_bitcnt > 0 && tlen) tmp[0] |= (u8)c->s_bitbuf;
        const size_t full = job.final ? tlen : (size_t)(bitpos / 8);
        if (!job.final) {
            c->s_bitcnt = (int)(bitpos & 7);
            c->s_bitbuf = c->s_bitcnt ? tmp[full] : 0;
        } else {
            c->s_bitcnt = 0;
            c->s_bitbuf = 0;
        }
        {
            std::lock_guard<std::mutex> lk(W.mu);
            if (full) {
                W.oq.push_back({ob, 0, full});
                W.oq_bytes += full;
            } else {
                W.out_busy[ob] = false;
            }
        }
    }
    if (c->debug)
        fprintf(stderr, "[bz2hip] stream job: %zu carried + %zu new bytes, %zu used, final %d, %.2f ms (set-up %.2f, compress %.2f)\n",
                W.carry, job.n, used, (int)job.final,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_job).count(),
                std::chrono::duration<double, std::milli>(t_a - t_job).count(),
                std::chrono::duration<double, std::milli>(t_b - t_a).count());
    // the bytes of the incomplete last block: into the gap in front of the other buffer's data, where
    // the next job (the buffers alternate) expects them
    const size_t rest = job.final ? 0 : n - used;
    if (rest > W.gap) throw HipError(BZ2HIP_E_HIP, "internal: an incomplete block holds more input than the gap in front of the stream buffer");
    if (rest) {
        HIPCHK(hipMemcpyAsync(W.d_buf[job.buf ^ 1] + W.gap - rest, d_input + used, rest, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    W.carry = rest;
}

static void stream_worker(bz2hip_ctx *c) {
    auto &W = c->sw;
    (void)hipSetDevice(c->device);
    for (;;) {
        bz2hip_ctx::StreamWorker::Job job;
        bool skip;
        {
            std::unique_lock<std::mutex> lk(W.mu);
            W.cv.wait(lk, [&] { return W.quit || !W.q.empty(); });
            if (W.q.empty()) return;
            job = W.q.front();
            W.q.pop_front();
            skip = W.err != 0; // after an error the rest of the stream is dropped
        }
        int rc = BZ2HIP_OK;
        std::string msg;
        if (!skip) {
            try {
                stream_job(c, job);
            } catch (const HipError &e) {
                rc = e.code;
                msg = e.what();
                (void)hipGetLastError();
            } catch (const std::bad_alloc &) {
                rc = BZ2HIP_E_NOMEM;
                msg = "host allocation failed";
            } catch (const std::exception &e) {
                rc = BZ2HIP_E_HIP;
                msg = e.what();
            }
        }
        {
            std::lock_guard<std::mutex> lk(W.mu);
            if (rc != BZ2HIP_OK && !W.err) {
                W.err = rc;
                W.errmsg = msg;
            }
            W.busy[job.buf] = false;
            W.active--;
        }
        W.cv.notify_all();
    }
}

// hands the buffer being filled to the worker and takes the other one (waits until it is free)
static void stream_submit(bz2hip_ctx *c, bool final) {
    auto &W = c->sw;
    std::unique_lock<std::mutex> lk(W.mu);
    if (!W.running) {
        W.quit = false;
        W.th = std::thread(stream_worker, c);
        W.running = true;
    }
    W.busy[W.cur] = true;
    W.first_job = false;
    // every piece of this buffer has been queued on the copy stream: the job waits for this event
    if (W.ev_up[W.cur]) (void)hipEventRecord(W.ev_up[W.cur], c->stream_copy);
    W.q.push_back({W.cur, W.fill, final});
    W.active++;
    W.cv.notify_all();
    W.cur ^= 1;
    W.fill = 0;
    W.cv.wait(lk, [&] { return !W.busy[W.cur]; });
}

static int stream_error(bz2hip_ctx *c) {
    std::lock_guard<std::mutex> lk(c->sw.mu);
    if (!c->sw.err) return BZ2HIP_OK;
    c->err = c->sw.errmsg;
    return c->sw.err;
}

enum { STREAM_GO_SYNC = 1 }; // stream_write_async: nothing taken, use the synchronous path (not an error code)
static int stream_write_async(bz2hip_ctx *c, const uint8_t *p, size_t n) {
    auto &W = c->sw;
    if (int rc = stream_error(c)) return rc;
    // Hand-over rule: a buffer holds two chunks; it goes to the worker when it is full, or as soon
    // as the worker is idle and half a chunk has gathered -- so the first batch starts early and
    // later ones are as large as the GPU's pace allows (large batches compress faster).
    const size_t cap = 2 * c->stream_chunk, piece = (size_t)4 << 20;
    // the first hand-over of a stream has a threshold of its own (BZ2HIP_STREAM_FIRST_MIB; measured
    // with a caller that writes at 20+ GB/s: 8 / 16 / 32 MiB give 37.4 / 37.5 / 35.9 ms per 256 MiB, so
    // it defaults to the later ones' 32 MiB); later ones are as large as the GPU's pace allows
    const size_t least_later = c->stream_least ? std::min(c->stream_least, cap) : c->stream_chunk / 2;
    const size_t least = W.first_job ? std::min(least_later, c->stream_first) : least_later;
    // device buffers behind a gap that holds whatever an incomplete block can carry over
    const size_t gap = ((size_t)c->s_level * 100000u * 52u + 65536u + 255u) & ~(size_t)255;
    if (hipSetDevice(c->device) != hipSuccess) return fail(c, BZ2HIP_E_HIP, "hipSetDevice failed");
    if (W.pin_cap == cap && (W.gap < gap || W.d_buf_cap < cap)) {
        // (no job is in flight and nothing is buffered: a stream of a higher level than the buffers were made for)
        for (int k = 0; k < 2; k++) {
            if (W.d_buf[k]) (void)hipFree(W.d_buf[k]);
            W.d_buf[k] = nullptr;
        }
        W.d_buf_cap = 0;
        W.pin_cap = 0; // sized again below
        for (int k = 0; k < 2; k++) {
            if (W.pin[k]) (void)hipHostFree(W.pin[k]);
            W.pin[k] = nullptr;
        }
    }
    if (W.pin_cap != cap) {
        // (no job is in flight here: the buffers are only sized between streams or at the first write)
        if (hipSetDevice(c->device) != hipSuccess) return fail(c, BZ2HIP_E_HIP, "hipSetDevice failed");
        for (int k = 0; k < 2; k++) {
            if (W.pin[k]) (void)hipHostFree(W.pin[k]);
            W.pin[k] = nullptr;
        }
        W.pin_cap = 0;
        for (int k = 0; k < 2; k++)
            if (hipHostMalloc((void **)&W.pin[k], cap, hipHostMallocDefault) != hipSuccess) {
                // no page-locked memory to be had: this stream compresses inside write / finish
                // (nothing has been buffered yet: the buffers are sized before the first byte is taken)
                (void)hipGetLastError();
                if (W.pin[0]) (void)hipHostFree(W.pin[0]);
                W.pin[0] = W.pin[1] = nullptr;
                c->s_async = false;
                return STREAM_GO_SYNC; // the caller goes on with the synchronous path
            }
        bool dev_ok = true;
        for (int k = 0; k < 2; k++) {
            if (W.d_buf[k]) (void)hipFree(W.d_buf[k]);
            W.d_buf[k] = nullptr;
            if (!W.ev_up[k] && hipEventCreateWithFlags(&W.ev_up[k], hipEventDisableTiming) != hipSuccess) dev_ok = false;
        }
        W.d_buf_cap = 0;
        for (int k = 0; k < 2 && dev_ok; k++)
            if (hipMalloc((void **)&W.d_buf[k], gap + cap + 64) != hipSuccess) dev_ok = false;
        if (!dev_ok) {
            // no device memory for the stream buffers: this stream compresses inside write / finish
            (void)hipGetLastError();
            for (int k = 0; k < 2; k++) {
                if (W.d_buf[k]) (void)hipFree(W.d_buf[k]);
                if (W.pin[k]) (void)hipHostFree(W.pin[k]);
                W.d_buf[k] = nullptr;
                W.pin[k] = nullptr;
            }
            c->s_async = false;
            return STREAM_GO_SYNC;
        }
        W.d_buf_cap = cap;
        W.gap = gap;
        W.pin_cap = cap;
    }
    while (n) {
        const size_t k = std::min(std::min(W.pin_cap - W.fill, n), piece);
        memcpy(W.pin[W.cur] + W.fill, p, k);
        // up it goes at once: by the time the buffer is handed over its bytes are in HBM
        if (hipMemcpyAsync(W.d_buf[W.cur] + W.gap + W.fill, W.pin[W.cur] + W.fill, k, hipMemcpyHostToDevice,
                           c->stream_copy) != hipSuccess) {
            (void)hipGetLastError();
            return fail(c, BZ2HIP_E_HIP, "upload of a stream chunk failed");
        }
        W.fill += k;
        p += k;
        n -= k;
        bool go = W.fill == W.pin_cap;
        if (!go && W.fill >= least) {
            std::lock_guard<std::mutex> lk(W.mu);
            go = W.active == 0;
        }
        if (go) {
            stream_submit(c, false);
            if (int rc = stream_error(c)) return rc; // the rest of the stream would be dropped anyway
        }
    }
    return stream_error(c);
}

extern "C" int bz2hip_stream_write(bz2hip_ctx *c, const uint8_t *p, size_t n) {
    if (!c) return BZ2HIP_E_PARAM;
    if (!c->s_open || c->s_finished) return fail(c, BZ2HIP_E_STATE, "Write beyond end of stream");
    if (n && !p) return fail(c, BZ2HIP_E_PARAM, "null pointer");
    if (c->s_async) {
        try {
            const int rc = stream_write_async(c, p, n);
            if (rc != STREAM_GO_SYNC) return rc;
        } catch (const std::exception &e) { // std::thread could not start
            return fail(c, BZ2HIP_E_NOMEM, e.what());
        }
    }
    try {
        c->s_in.insert(c->s_in.end(), p, p + n);
    } catch (const std::bad_alloc &) {
        return fail(c, BZ2HIP_E_NOMEM, "host allocation failed");
    }
    if (c->s_in.size() - c->s_in_pos >= c->stream_chunk * (c->peers.size() + 1)) return stream_flush(c, false);
    return BZ2HIP_OK;
}

extern "C" int bz2hip_stream_finish(bz2hip_ctx *c) {
    if (!c) return BZ2HIP_E_PARAM;
    if (!c->s_open) return fail(c, BZ2HIP_E_STATE, "stream not open");
    if (c->s_finished) return BZ2HIP_OK; // finish() is idempotent (S/BZip2OutputStream.swift:151)
    if (c->s_async) {
        if (int rc0 = stream_error(c)) return rc0;
        try {
            stream_submit(c, true);
        } catch (const std::exception &e) { // std::thread could not start
            return fail(c, BZ2HIP_E_NOMEM, e.what());
        }
        stream_drain(c);
        const int rc1 = stream_error(c);
        if (rc1 == BZ2HIP_OK) c->s_finished = true;
        return rc1;
    }
    int rc = stream_flush(c, true);
    if (rc == BZ2HIP_OK) c->s_finished = true;
    return rc;
}

extern "C" int bz2hip_stream_read(bz2hip_ctx *c, uint8_t *out, size_t cap, size_t *produced) {
    if (!c || !produced) return BZ2HIP_E_PARAM;
    auto &W = c->sw;
    std::lock_guard<std::mutex> lk(W.mu);
    size_t avail = c->s_out.size() - c->s_out_pos;
    size_t k = std::min(avail, cap);
    if (k) memcpy(out, c->s_out.data() + c->s_out_pos, k);
    c->s_out_pos += k;
    if (c->s_out_pos == c->s_out.size()) {
        c->s_out.clear();
        c->s_out_pos = 0;
    }
    // then the jobs' landing buffers, in stream order
    while (k < cap && c->s_out_pos == c->s_out.size() && !W.oq.empty()) {
        auto &sg = W.oq.front();
        const size_t m = std::min(sg.len, cap - k);
        memcpy(out + k, W.pin_out[sg.buf] + sg.off, m);
        k += m;
        sg.off += m;
        sg.len -= m;
        W.oq_bytes -= m;
        if (sg.len == 0) {
            const int b = sg.buf;
            W.oq.pop_front();
            bool more = false;
            for (const auto &o : W.oq) more = more || o.buf == b;
            if (!more) W.out_busy[b] = false;
        }
    }
    *produced = k;
    return BZ2HIP_OK;
}

extern "C" size_t bz2hip_stream_pending(const bz2hip_ctx *c) {
    if (!c) return 0;
    std::lock_guard<std::mutex> lk(const_cast<bz2hip_ctx *>(c)->sw.mu);
    return c->s_out.size() - c->s_out_pos + c->sw.oq_bytes;
}

// ------------------------------------------------------------------ stage entry points
static void upload_present(bz2hip_ctx *c, const uint8_t *present, u32 b0, int nb) {
    std::vector<u32> w((size_t)nb * 8, 0);
    for (int b = 0; b < nb; b++)
        for (int v = 0; v < 256; v++)
            if (present[((size_t)b0 + b) * 256 + v]) w[(size_t)b * 8 + (v >> 5)] |= 1u << (v & 31);
    HIPCHK(hipMemcpy(c->batch.present, w.data(), w.size() * sizeof(u32), hipMemcpyHostToDevice));
}

static int level_for_maxn(u32 maxn) {
    int level = (int)((maxn + 99999) / 100000);
    if (level < 1) level = 1;
    return level;
}

// declared block size (what the reference's caller sized `SA` for: blockSize + 1 ints,
// S/BZip2BlockCompressor.swift:318) -> level; 0 = the smallest level that holds the longest block
static int level_for_declared(u32 declared, u32 maxn, int *level) {
    if (declared == 0) { *level = level_for_maxn(maxn); return BZ2HIP_OK; }
    if (declared % 100000u != 0 || declared < 100000u || declared > 900000u || maxn > declared) return BZ2HIP_E_PARAM;
    *level = (int)(declared / 100000u);
    return BZ2HIP_OK;
}

extern "C" int bz2hip_bwt(bz2hip_ctx *c, const uint8_t *blocks, const uint32_t *n, uint32_t count,
                          uint32_t stride, uint8_t *bwt, uint32_t *orig, uint32_t *rounds,
                          uint32_t declared_block_size) {
    return guarded(c, [&]() -> int {
        if (!blocks || !n || !bwt || !orig) return fail(c, BZ2HIP_E_PARAM, "null pointer");
        u32 maxn = 0;
        for (u32 b = 0; b < count; b++) maxn = std::max(maxn, n[b]);
        if (maxn > 900000 || maxn > stride) return fail(c, BZ2HIP_E_PARAM, "block longer than 900000 or stride");
        int blevel = 0;
        if (level_for_declared(declared_block_size, maxn, &blevel) != BZ2HIP_OK)
            return fail(c, BZ2HIP_E_PARAM, "declared_block_size must be 0 or level * 100000 and hold every block");
        ensure_batch(c, blevel, (int)count);
        Batch &B = c->batch;
        for (u32 b0 = 0; b0 < count; b0 += (u32)B.NB) {
            int nb = (int)std::min<u32>((u32)B.NB, count - b0);
            HIPCHK(hipMemcpy2D(B.T, B.S, blocks + (size_t)b0 * stride, stride, maxn, nb, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(B.n, n + b0, nb * sizeof(u32), hipMemcpyHostToDevice));
            run_present(c, nb, maxn);
            run_bwt_n(c, nb, maxn);
            run_tie_finish(c, nb);
            HIPCHK(hipStreamSynchronize(c->stream));
            if (maxn) HIPCHK(hipMemcpy2D(bwt + (size_t)b0 * stride, stride, B.bwt, B.S, maxn, nb, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(orig + b0, B.orig, nb * sizeof(u32), hipMemcpyDeviceToHost));
            if (rounds) HIPCHK(hipMemcpy(rounds + b0, B.rounds, nb * sizeof(u32), hipMemcpyDeviceToHost));
        }
        if (c->profile) c->prof_collect();
        return BZ2HIP_OK;
    });
}

extern "C" int bz2hip_mtf_rle2(bz2hip_ctx *c, const uint8_t *bwt, const uint32_t *n, uint32_t count,
                               uint32_t stride, const uint8_t *present, uint16_t *mtf,
                               uint32_t *mtf_len, uint32_t *alpha, int32_t *freq) {
    return guarded(c, [&]() -> int {
        if (!bwt || !n || !present || !mtf || !mtf_len || !alpha || !freq)
            return fail(c, BZ2HIP_E_PARAM, "null pointer");
        u32 maxn = 0;
        for (u32 b = 0; b < count; b++) maxn = std::max(maxn, n[b]);
        if (maxn > 900000 || maxn > stride) return fail(c, BZ2HIP_E_PARAM, "block longer than 900000 or stride");
        ensure_batch(c, level_for_maxn(maxn), (int)count);
        Batch &B = c->batch;
        const size_t S2 = B.S + 64;
        for (u32 b0 = 0; b0 < count; b0 += (u32)B.NB) {
            int nb = (int)std::min<u32>((u32)B.NB, count - b0);
            if (maxn) HIPCHK(hipMemcpy2D(B.bwt, B.S, bwt + (size_t)b0 * stride, stride, maxn, nb, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(B.n, n + b0, nb * sizeof(u32), hipMemcpyHostToDevice));
            upload_present(c, present, b0, nb);
            run_mtf(c, nb, maxn);
            HIPCHK(hipStreamSynchronize(c->stream));
            HIPCHK(hipMemcpy2D(mtf + (size_t)b0 * (stride + 2), (size_t)(stride + 2) * 2, B.mtf, S2 * 2,
                               (size_t)(maxn + 1) * 2, nb, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(mtf_len + b0, B.mtflen, nb * sizeof(u32), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(alpha + b0, B.alpha, nb * sizeof(u32), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy2D(freq + (size_t)b0 * 258, 258 * 4, B.freq, 264 * 4, 258 * 4, nb, hipMemcpyDeviceToHost));
        }
        if (c->profile) c->prof_collect();
        return BZ2HIP_OK;
    });
}

extern "C" int bz2hip_huffman(bz2hip_ctx *c, const uint16_t *mtf, const uint32_t *mtf_len,
                              const uint32_t *alpha, const int32_t *freq, uint32_t count,
                              uint32_t stride, uint32_t *n_tables, int32_t *lengths,
                              uint8_t *selectors, uint32_t sel_stride, uint32_t *n_selectors) {
    return guarded(c, [&]() -> int {
        if (!mtf || !mtf_len || !alpha || !freq || !n_tables || !lengths || !selectors || !n_selectors)
            return fail(c, BZ2HIP_E_PARAM, "null pointer");
        u32 maxl = 0;
        for (u32 b = 0; b < count; b++) maxl = std::max(maxl, mtf_len[b]);
        if (maxl > 900001 || maxl > stride) return fail(c, BZ2HIP_E_PARAM, "mtf block too long");
        ensure_batch(c, level_for_maxn(maxl > 0 ? maxl - 1 : 0), (int)count);
        Batch &B = c->batch;
        const size_t S2 = B.S + 64;
        for (u32 b0 = 0; b0 < count; b0 += (u32)B.NB) {
            int nb = (int)std::min<u32>((u32)B.NB, count - b0);
            HIPCHK(hipMemcpy2D(B.mtf, S2 * 2, mtf + (size_t)b0 * stride, (size_t)stride * 2, (size_t)maxl * 2, nb,
                               hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(B.mtflen, mtf_len + b0, nb * sizeof(u32), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(B.alpha, alpha + b0, nb * sizeof(u32), hipMemcpyHostToDevice));
            HIPCHK(hipMemset(B.freq, 0, (size_t)nb * 264 * 4));
            HIPCHK(hipMemcpy2D(B.freq, 264 * 4, freq + (size_t)b0 * 258, 258 * 4, 258 * 4, nb, hipMemcpyHostToDevice));
            run_huffman(c, nb, maxl);
            HIPCHK(hipStreamSynchronize(c->stream));
            HIPCHK(hipMemcpy(n_tables + b0, B.ntables, nb * sizeof(u32), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(n_selectors + b0, B.nsel, nb * sizeof(u32), hipMemcpyDeviceToHost));
            std::vector<u8> hl((size_t)nb * MAX_TABLES * 264);
            HIPCHK(hipMemcpy(hl.data(), B.hlen, hl.size(), hipMemcpyDeviceToHost));
            for (int b = 0; b < nb; b++)
                for (int t = 0; t < MAX_TABLES; t++)
                    for (int j = 0; j < 258; j++)
                        lengths[(((size_t)b0 + b) * MAX_TABLES + t) * 258 + j] =
                            hl[((size_t)b * MAX_TABLES + t) * 264 + j];
            u32 w = std::min<u32>(sel_stride, SEL_STRIDE);
            HIPCHK(hipMemcpy2D(selectors + (size_t)b0 * sel_stride, sel_stride, B.sel, SEL_STRIDE, w, nb,
                               hipMemcpyDeviceToHost));
        }
        if (c->profile) c->prof_collect();
        return BZ2HIP_OK;
    });
}

extern "C" int bz2hip_block_compress(bz2hip_ctx *c, const uint8_t *rle_blocks, const uint32_t *n,
                                     uint32_t count, uint32_t stride, const uint8_t *present,
                                     const uint32_t *crc, uint8_t *bits, size_t bits_stride,
                                     uint64_t *nbits, uint32_t declared_block_size) {
    return guarded(c, [&]() -> int {
        if (!rle_blocks || !n || !present || !crc || !bits || !nbits)
            return fail(c, BZ2HIP_E_PARAM, "null pointer");
        if (bits_stride & 3) return fail(c, BZ2HIP_E_PARAM, "bits_stride must be a multiple of 4");
        u32 maxn = 0;
        for (u32 b = 0; b < count; b++) {
            if (n[b] == 0) return fail(c, BZ2HIP_E_PARAM, "empty block");
            maxn = std::max(maxn, n[b]);
        }
        if (maxn > 900000 || maxn > stride) return fail(c, BZ2HIP_E_PARAM, "block longer than 900000 or stride");
        int blevel = 0;
        if (level_for_declared(declared_block_size, maxn, &blevel) != BZ2HIP_OK)
            return fail(c, BZ2HIP_E_PARAM, "declared_block_size must be 0 or level * 100000 and hold every block");
        ensure_batch(c, blevel, (int)count);
        Batch &B = c->batch;
        ensure_staging(c, 0, (size_t)B.NB * bits_stride);
        for (u32 b0 = 0; b0 < count; b0 += (u32)B.NB) {
            int nb = (int)std::min<u32>((u32)B.NB, count - b0);
            HIPCHK(hipMemcpy2D(B.T, B.S, rle_blocks + (size_t)b0 * stride, stride, maxn, nb, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(B.n, n + b0, nb * sizeof(u32), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(B.crc, crc + b0, nb * sizeof(u32), hipMemcpyHostToDevice));
            upload_present(c, present, b0, nb);
            HIPCHK(hipMemsetAsync(c->h_out, 0, (size_t)nb * bits_stride, c->stream));
            run_bwt_n(c, nb, maxn);
            run_mtf(c, nb, maxn);
            run_huffman(c, nb, maxn + 1);
            std::vector<u64> hb;
            run_pack(c, nb, maxn + 1, c->h_out, 0, true, bits_stride, 0, &hb);
            HIPCHK(hipStreamSynchronize(c->stream));
            HIPCHK(hipMemcpy(bits + (size_t)b0 * bits_stride, c->h_out, (size_t)nb * bits_stride, hipMemcpyDeviceToHost));
            for (int b = 0; b < nb; b++) nbits[b0 + b] = hb[b];
        }
        if (c->profile) c->prof_collect();
        return BZ2HIP_OK;
    });
}

extern "C" int bz2hip_rle1_split(bz2hip_ctx *c, const uint8_t *in, size_t n, int level,
                                 uint32_t max_blocks, uint32_t *n_blocks, uint64_t *starts,
                                 uint8_t *rle, uint32_t *rle_len, uint8_t *present, uint32_t *crc) {
    return guarded(c, [&]() -> int {
        if (level < 1 || level > 9) return fail(c, BZ2HIP_E_PARAM, "Invalid BZip2 block size");
        if (!n_blocks || !starts || (n && !in)) return fail(c, BZ2HIP_E_PARAM, "null pointer");
        *n_blocks = 0;
        if (n == 0) return BZ2HIP_OK;
        const u32 bs = (u32)level * 100000u;
        const size_t rstride = (size_t)bs + 8;
        const bool emit = rle && rle_len && present && crc;
        // windows of at most WMAX bytes, as the whole-stream path cuts them: a block that ends at
        // the end of a window that is not the last is dropped there and begins the next window
        size_t pos = 0;
        u32 total = 0;
        std::vector<u32> hs;
        while (pos < n) {
            const size_t wlen = std::min<size_t>(n - pos, WMAX);
            const bool last_window = (pos + wlen == n);
            ensure_staging(c, wlen, 0);
            HIPCHK(hipMemcpyAsync(c->h_in, in + pos, wlen, hipMemcpyHostToDevice, c->stream));
            prepare_window(c, c->h_in, wlen, level);
            u32 cnt = run_chain(c, c->h_in, (u32)wlen, level, c->w_rs, c->w_P, c->w_agg2, c->w_starts,
                                c->w_maxblocks, 0, (u32)wlen, hs);
            u32 usable = cnt;
            if (!last_window && cnt > 0 && hs[cnt] == (u32)wlen) usable = cnt - 1;
            if (usable == 0) throw HipError(BZ2HIP_E_HIP, "window holds no complete block");
            if (total + usable > max_blocks) return fail(c, BZ2HIP_E_CAP, "more blocks than max_blocks");
            for (u32 b = 0; b < usable; b++) starts[total + b] = pos + hs[b];
            if (emit) {
                Batch &B = c->batch;
                for (u32 b0 = 0; b0 < usable; b0 += (u32)B.NB) {
                    int nb = (int)std::min<u32>((u32)B.NB, usable - b0);
                    u32 max_in = 0;
                    for (int b = 0; b < nb; b++) max_in = std::max(max_in, hs[b0 + b + 1] - hs[b0 + b]);
                    run_rle1_emit(c, c->h_in, c->w_agg1, c->w_starts, b0, nb, max_in, c->w_ETPB);
                    HIPCHK(hipStreamSynchronize(c->stream));
                    const size_t o = (size_t)total + b0;
                    HIPCHK(hipMemcpy2D(rle + o * rstride, rstride, B.T, B.S, bs, nb, hipMemcpyDeviceToHost));
                    HIPCHK(hipMemcpy(rle_len + o, B.n, nb * sizeof(u32), hipMemcpyDeviceToHost));
                    HIPCHK(hipMemcpy(crc + o, B.crc, nb * sizeof(u32), hipMemcpyDeviceToHost));
                    std::vector<u32> pw((size_t)nb * 8);
                    HIPCHK(hipMemcpy(pw.data(), B.present, pw.size() * 4, hipMemcpyDeviceToHost));
                    for (int b = 0; b < nb; b++)
                        for (int v = 0; v < 256; v++)
                            present[(o + b) * 256 + v] = (pw[(size_t)b * 8 + (v >> 5)] >> (v & 31)) & 1u;
                }
            }
            total += usable;
            pos += hs[usable];
        }
        *n_blocks = total;
        if (c->profile) c->prof_collect();
        return BZ2HIP_OK;
    });
}

// ------------------------------------------------------------------ profiling
static const char *k_prof_names[P_COUNT] = {
    "rle1_scan", "rle1_split", "rle1_emit", "crc", "bwt_init", "radix_hist", "radix_scan",
    "radix_scatter", "bwt_keys", "bwt_ssort", "bwt_rerank", "bwt_emit", "bwt_tie", "mtf_state", "mtf_compose",
    "mtf_replay", "rle2", "huf_seed", "huf_select", "huf_lengths", "huf_codes", "huf_header",
    "pack", "frame"};

extern "C" int bz2hip_profile_reset(bz2hip_ctx *c) {
    if (!c) return BZ2HIP_E_PARAM;
    for (int i = 0; i < P_COUNT; i++) {
        c->prof_ms[i] = 0;
        c->prof_launches[i] = 0;
        c->prof_bytes[i] = 0;
    }
    return BZ2HIP_OK;
}

extern "C" int bz2hip_profile_get(bz2hip_ctx *c, int *count, const char **names, double *ms,
                                  uint64_t *launches, double *alg_bytes) {
    if (!c || !count) return BZ2HIP_E_PARAM;
    *count = P_COUNT;
    for (int i = 0; i < P_COUNT; i++) {
        if (names) names[i] = k_prof_names[i];
        if (ms) ms[i] = c->prof_ms[i];
        if (launches) launches[i] = c->prof_launches[i];
        if (alg_bytes) alg_bytes[i] = c->prof_bytes[i];
    }
    return BZ2HIP_OK;
}

extern "C" int bz2hip_last_stats2(bz2hip_ctx *c, uint64_t *n_times_rounds, uint64_t *sorted_elems) {
    if (!c) return BZ2HIP_E_PARAM;
    if (n_times_rounds) *n_times_rounds = c->st_nr;
    if (sorted_elems) *sorted_elems = c->st_sorted_elems;
    return BZ2HIP_OK;
}

extern "C" int bz2hip_last_stats3(bz2hip_ctx *c, double *bwt_seconds, uint64_t *tie_traps) {
    if (!c) return BZ2HIP_E_PARAM;
    if (bwt_seconds) *bwt_seconds = c->st_bwt_seconds;
    if (tie_traps) *tie_traps = c->st_tie_traps;
    return BZ2HIP_OK;
}

extern "C" int bz2hip_decode_consumed(bz2hip_ctx *c, size_t *consumed) {
    if (!c || !consumed) return BZ2HIP_E_PARAM;
    *consumed = c->dec_consumed;
    return BZ2HIP_OK;
}

extern "C" int bz2hip_tie_stats(bz2hip_ctx *c, uint64_t *replays, uint64_t *reused, uint64_t *cached_keys) {
    if (!c) return BZ2HIP_E_PARAM;
    u64 r = c->st_tie_replays, u = c->st_tie_cached, k = c->tie_cache.size();
    for (bz2hip_ctx *p : c->peers) { r += p->st_tie_replays; u += p->st_tie_cached; k += p->tie_cache.size(); }
    if (replays) *replays = r;
    if (reused) *reused = u;
    if (cached_keys) *cached_keys = k;
    return BZ2HIP_OK;
}

extern "C" int bz2hip_tie_stats2(bz2hip_ctx *c, uint64_t *rejected) {
    if (!c) return BZ2HIP_E_PARAM;
    u64 r = c->st_tie_rejected;
    for (bz2hip_ctx *p : c->peers) r += p->st_tie_rejected;
    if (rejected) *rejected = r;
    return BZ2HIP_OK;
}

extern "C" int bz2hip_last_stats(bz2hip_ctx *c, uint64_t *blocks, uint64_t *rounds_sum,
                                 uint64_t *rle_bytes, uint64_t *periodic_blocks) {
    if (!c) return BZ2HIP_E_PARAM;
    if (blocks) *blocks = c->st_blocks;
    if (rounds_sum) *rounds_sum = c->st_rounds;
    if (rle_bytes) *rle_bytes = c->st_rle_bytes;
    if (periodic_blocks) *periodic_blocks = c->st_periodic;
    return BZ2HIP_OK;
}
