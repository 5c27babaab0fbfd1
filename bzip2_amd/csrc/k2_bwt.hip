// k2_bwt.hip -- batched cyclic-rotation sort (BWT + origin pointer) by prefix doubling.
//
// Replaces BZip2DivSufSort.bwt() (S/BZip2DivSufSort.swift:3487-3506, sortTypeBstar :3076-3362,
// constructBWT :3369-3481).  The BWT string of a block is unique, so any correct rotation sort
// reproduces it; `orig` is unique unless the block is a perfect power w^k.
//
// Algorithm (all blocks of the batch in every launch, grid.y = block):
//   round 0 : key64 = first 8 bytes of each rotation (cyclic), 8 LSD passes of 8 bits
//   round r : key40 = (rank(i) << 20) | rank((i+h) mod n), 5 LSD passes, h = 8,16,32,...
//   after every sort: head flags -> rank = position of the group's head (max-scan)
//   a block stops when every group is a singleton, or when h >= n (then it is periodic).
// Each LSD pass = tile digit histogram (LDS atomics) -> per-block scan -> stable scatter
// (wave64 ballot matching, per-wave LDS counters).  HBM-bound; no MFMA.
#include "ctx.hpp"

// ------------------------------------------------------------------ round 0 keys
__global__ __launch_bounds__(RT_THREADS) void k2_init_keys(const u8 *__restrict__ T,
                                                           const u32 *__restrict__ nArr,
                                                           u64 *__restrict__ key,
                                                           u32 *__restrict__ val, u32 S) {
    const u32 b = blockIdx.y;
    const u32 n = nArr[b];
    const u32 base = blockIdx.x * RTILE;
    if (base >= n) return;
    const u8 *t = T + (size_t)b * S;
#pragma unroll 4
    for (int k = 0; k < RT_ITEMS; k++) {
        u32 i = base + k * RT_THREADS + threadIdx.x;
        if (i >= n) break;
        u64 kk = 0;
        if (i + 8 <= n) {
#pragma unroll
            for (int q = 0; q < 8; q++) kk = (kk << 8) | t[i + q];
        } else {
            u32 p = i;
            for (int q = 0; q < 8; q++) {
                kk = (kk << 8) | t[p];
                p++;
                if (p >= n) p = 0;
            }
        }
        key[(size_t)b * S + i] = kk;
        val[(size_t)b * S + i] = i;
    }
}

// ------------------------------------------------------------------ LSD radix pass
__global__ __launch_bounds__(RT_THREADS) void k2_radix_hist(const u64 *__restrict__ key,
                                                            const u32 *__restrict__ nArr,
                                                            const u32 *__restrict__ active,
                                                            u32 *__restrict__ tilehist, u32 S,
                                                            u32 TPB, int shift) {
    const u32 b = blockIdx.y;
    if (!active[b]) return;
    const u32 n = nArr[b];
    const u32 base = blockIdx.x * RTILE;
    if (base >= n) return;
    __shared__ u32 hist[256];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const u64 *kp = key + (size_t)b * S;
#pragma unroll
    for (int k = 0; k < RT_ITEMS; k++) {
        u32 i = base + k * RT_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&hist[(u32)(kp[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    tilehist[((size_t)b * TPB + blockIdx.x) * 256 + threadIdx.x] = hist[threadIdx.x];
}

// one workgroup per block: exclusive scan of tile counts per digit, then digit bases
__global__ __launch_bounds__(256) void k2_radix_scan(const u32 *__restrict__ nArr,
                                                     const u32 *__restrict__ active,
                                                     u32 *__restrict__ tilehist,
                                                     u32 *__restrict__ digitbase, u32 TPB) {
    const u32 b = blockIdx.x;
    if (!active[b]) return;
    const u32 n = nArr[b];
    const u32 tiles = (n + RTILE - 1) / RTILE;
    __shared__ u32 sm[8];
    u32 *th = tilehist + (size_t)b * TPB * 256 + threadIdx.x;
    u32 run = 0;
    for (u32 t = 0; t < tiles; t++) {
        u32 c = th[(size_t)t * 256];
        th[(size_t)t * 256] = run;
        run += c;
    }
    u32 total;
    u32 ex = block_excl_scan_add<256>(run, sm, &total);
    digitbase[b * 256 + threadIdx.x] = ex;
}

__device__ __forceinline__ u64 match8(u32 d, bool valid) {
    u64 m = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; bit++) {
        bool s = (d >> bit) & 1;
        u64 bb = __ballot(s);
        m &= s ? bb : ~bb;
    }
    return m;
}

__global__ __launch_bounds__(RT_THREADS) void k2_radix_scatter(
    const u64 *__restrict__ keyIn, const u32 *__restrict__ valIn, u64 *__restrict__ keyOut,
    u32 *__restrict__ valOut, const u32 *__restrict__ nArr, const u32 *__restrict__ active,
    const u32 *__restrict__ tilehist, const u32 *__restrict__ digitbase, u32 S, u32 TPB,
    int shift) {
    const u32 b = blockIdx.y;
    if (!active[b]) return;
    const u32 n = nArr[b];
    const u32 base = blockIdx.x * RTILE;
    if (base >= n) return;
    __shared__ u32 whist[4][256];
    __shared__ u32 gofs[256];
    const int w = wave_id(), lane = lane_id();
    for (int q = threadIdx.x; q < 4 * 256; q += RT_THREADS) (&whist[0][0])[q] = 0;
    __syncthreads();
    const u64 *kp = keyIn + (size_t)b * S;
    const u32 *vp = valIn + (size_t)b * S;
    u64 keys[RT_ITEMS];
    u32 vals[RT_ITEMS];
    u32 lr[RT_ITEMS];
    const u64 lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < RT_ITEMS; k++) {
        u32 i = base + w * (RTILE / 4) + k * 64 + lane;
        bool valid = i < n;
        u64 kk = valid ? kp[i] : 0;
        u32 vv = valid ? vp[i] : 0;
        keys[k] = kk;
        vals[k] = vv;
        u32 d = (u32)(kk >> shift) & 255u;
        u64 m = match8(d, valid);
        u32 r = __popcll(m & lt);
        u32 cnt = __popcll(m);
        int leader = __ffsll((unsigned long long)m) - 1;
        u32 old = 0;
        if (valid && lane == leader) {
            old = whist[w][d];
            whist[w][d] = old + cnt;
        }
        old = __shfl(old, leader < 0 ? 0 : leader, 64);
        lr[k] = old + r;
    }
    __syncthreads();
    {
        // exclusive prefix over waves per digit + global offset of this tile's digit run
        const u32 d = threadIdx.x;
        u32 run = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            u32 c = whist[q][d];
            whist[q][d] = run;
            run += c;
        }
        gofs[d] = digitbase[b * 256 + d] + tilehist[((size_t)b * TPB + blockIdx.x) * 256 + d];
    }
    __syncthreads();
    u64 *ko = keyOut + (size_t)b * S;
    u32 *vo = valOut + (size_t)b * S;
#pragma unroll
    for (int k = 0; k < RT_ITEMS; k++) {
        u32 i = base + w * (RTILE / 4) + k * 64 + lane;
        if (i < n) {
            u32 d = (u32)(keys[k] >> shift) & 255u;
            u32 dst = gofs[d] + whist[w][d] + lr[k];
            ko[dst] = keys[k];
            vo[dst] = vals[k];
        }
    }
}

// ------------------------------------------------------------------ re-rank
// head(j) = j == 0 || key[j] != key[j-1];  grp[j] = position of nearest head <= j.
// phase 1: per tile, last head position + 1 (0 = none) and head count.
__global__ __launch_bounds__(RT_THREADS) void k2_rerank_p1(const u64 *__restrict__ key,
                                                           const u32 *__restrict__ nArr,
                                                           const u32 *__restrict__ active,
                                                           u32 *__restrict__ tileagg,
                                                           u32 *__restrict__ nheads, u32 S,
                                                           u32 TPB) {
    const u32 b = blockIdx.y;
    if (!active[b]) return;
    const u32 n = nArr[b];
    const u32 base = blockIdx.x * RTILE;
    if (base >= n) return;
    __shared__ u32 sm[8];
    const u64 *kp = key + (size_t)b * S;
    u32 last = 0, cnt = 0;
#pragma unroll 4
    for (int k = 0; k < RT_ITEMS; k++) {
        u32 j = base + k * RT_THREADS + threadIdx.x;
        if (j < n) {
            bool head = (j == 0) || (kp[j] != kp[j - 1]);
            if (head) {
                last = j + 1;
                cnt++;
            }
        }
    }
    u32 tl = block_reduce_max<RT_THREADS>(last, sm);
    u32 tc = block_reduce_add<RT_THREADS>(cnt, sm);
    if (threadIdx.x == 0) {
        tileagg[(size_t)b * TPB + blockIdx.x] = tl;
        atomicAdd(&nheads[b], tc);
    }
}

// phase 2: exclusive max-scan of tile aggregates (one workgroup per block)
__global__ __launch_bounds__(256) void k2_rerank_p2(const u32 *__restrict__ nArr,
                                                    const u32 *__restrict__ active,
                                                    u32 *__restrict__ tileagg, u32 TPB) {
    const u32 b = blockIdx.x;
    if (!active[b]) return;
    const u32 tiles = (nArr[b] + RTILE - 1) / RTILE;
    __shared__ u32 sm[8];
    u32 carry = 0;
    for (u32 t0 = 0; t0 < tiles; t0 += 256) {
        u32 t = t0 + threadIdx.x;
        u32 v = (t < tiles) ? tileagg[(size_t)b * TPB + t] : 0;
        u32 tot;
        u32 inc = block_incl_scan_max<256>(v, sm, &tot);
        // exclusive = max over previous elements
        u32 prev = __shfl_up(inc, 1, 64);
        u32 ex;
        if (lane_id() == 0) {
            // previous wave's inclusive total
            ex = 0;
            for (int q = 0; q < wave_id(); q++) ex = max(ex, sm[q]);
        } else {
            ex = prev;
        }
        ex = max(ex, carry);
        if (t < tiles) tileagg[(size_t)b * TPB + t] = ex;
        carry = max(carry, tot);
        __syncthreads();
    }
}

// phase 3: grp[j], rank[sa[j]] = grp[j], sa[j] = val[j]
__global__ __launch_bounds__(RT_THREADS) void k2_rerank_p3(
    const u64 *__restrict__ key, const u32 *__restrict__ val, const u32 *__restrict__ nArr,
    const u32 *__restrict__ active, const u32 *__restrict__ tileagg, u32 *__restrict__ grp,
    u32 *__restrict__ rank, u32 *__restrict__ sa, u32 S, u32 TPB) {
    const u32 b = blockIdx.y;
    if (!active[b]) return;
    const u32 n = nArr[b];
    const u32 base = blockIdx.x * RTILE;
    if (base >= n) return;
    __shared__ u32 wlast[4];
    const int w = wave_id(), lane = lane_id();
    const u64 *kp = key + (size_t)b * S;
    const u32 *vp = val + (size_t)b * S;
    u64 hm[RT_ITEMS];
    u32 myv[RT_ITEMS];
    u32 last = 0; // last head (+1) seen by this wave
#pragma unroll
    for (int k = 0; k < RT_ITEMS; k++) {
        u32 j = base + w * (RTILE / 4) + k * 64 + lane;
        bool head = false;
        u32 v = 0;
        if (j < n) {
            head = (j == 0) || (kp[j] != kp[j - 1]);
            v = vp[j];
        }
        myv[k] = v;
        u64 m = __ballot(head);
        hm[k] = m;
        if (m) last = base + w * (RTILE / 4) + k * 64 + (63 - __clzll((unsigned long long)m)) + 1;
    }
    if (lane == 0) wlast[w] = last;
    __syncthreads();
    u32 carry = tileagg[(size_t)b * TPB + blockIdx.x];
    for (int q = 0; q < w; q++) carry = max(carry, wlast[q]);
    const u64 le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
#pragma unroll
    for (int k = 0; k < RT_ITEMS; k++) {
        u32 rowbase = base + w * (RTILE / 4) + k * 64;
        u32 j = rowbase + lane;
        u64 m = hm[k] & le;
        u32 g = m ? (rowbase + (63 - __clzll((unsigned long long)m)) + 1) : carry;
        if (hm[k]) carry = rowbase + (63 - __clzll((unsigned long long)hm[k])) + 1;
        if (j < n) {
            u32 gg = g - 1;
            grp[(size_t)b * S + j] = gg;
            sa[(size_t)b * S + j] = myv[k];
            rank[(size_t)b * S + myv[k]] = gg;
        }
    }
}

// per-round control: decide which blocks continue; one thread per block
__global__ void k2_round_ctl(const u32 *__restrict__ nArr, u32 *__restrict__ nheads,
                             u32 *__restrict__ active, u32 *__restrict__ rounds,
                             u32 *__restrict__ periodic, u32 *__restrict__ ctl, u32 nb,
                             u32 h_next) {
    u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    if (!active[b]) return;
    u32 n = nArr[b], heads = nheads[b];
    bool cont = (heads < n) && (h_next < n);
    if (cont) {
        rounds[b] += 1;
        nheads[b] = 0;
        atomicOr(&ctl[0], 1u);
        atomicAdd(&ctl[1], n); // elements sorted in the coming round (for the roofline figure)
    } else {
        active[b] = 0;
        periodic[b] = (heads < n) ? (n / heads) : 1;
    }
}

// ------------------------------------------------------------------ doubling keys
__global__ __launch_bounds__(RT_THREADS) void k2_build_keys(
    const u32 *__restrict__ sa, const u32 *__restrict__ grp, const u32 *__restrict__ rank,
    const u32 *__restrict__ nArr, const u32 *__restrict__ active, u64 *__restrict__ key,
    u32 *__restrict__ val, u32 S, u32 h) {
    const u32 b = blockIdx.y;
    if (!active[b]) return;
    const u32 n = nArr[b];
    const u32 base = blockIdx.x * RTILE;
    if (base >= n) return;
    const size_t o = (size_t)b * S;
#pragma unroll 4
    for (int k = 0; k < RT_ITEMS; k++) {
        u32 j = base + k * RT_THREADS + threadIdx.x;
        if (j < n) {
            u32 i = sa[o + j];
            u32 i2 = i + h;
            if (i2 >= n) i2 -= n;
            key[o + j] = ((u64)grp[o + j] << 20) | (u64)rank[o + i2];
            val[o + j] = i;
        }
    }
}

// ------------------------------------------------------------------ output
__global__ __launch_bounds__(RT_THREADS) void k2_emit_bwt(
    const u8 *__restrict__ T, const u32 *__restrict__ sa, const u32 *__restrict__ rank,
    const u32 *__restrict__ nArr, const u32 *__restrict__ periodic, u8 *__restrict__ bwt,
    u32 *__restrict__ orig, u32 S, int tie_rule) {
    const u32 b = blockIdx.y;
    const u32 n = nArr[b];
    const u32 base = blockIdx.x * RTILE;
    if (base >= n) return;
    const size_t o = (size_t)b * S;
#pragma unroll 4
    for (int k = 0; k < RT_ITEMS; k++) {
        u32 j = base + k * RT_THREADS + threadIdx.x;
        if (j < n) {
            u32 i = sa[o + j];
            i = (i == 0) ? n - 1 : i - 1;
            bwt[o + j] = T[o + i];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        u32 kper = periodic[b];
        u32 og = rank[o];
        if (kper > 1 && kper < n && tie_rule == 1) og += kper - 1;
        if (kper == n) og = 0; // all-equal block (S/BZip2DivSufSort.swift:3168-3174,3505)
        orig[b] = og;
    }
}

__global__ void k2_reset(const u32 *__restrict__ nArr, u32 *nheads, u32 *active, u32 *rounds,
                         u32 *periodic, u32 *ctl, u32 nb) {
    u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    atomicAdd(&ctl[2], nArr[b]);
    nheads[b] = 0;
    active[b] = nArr[b] > 0 ? 1 : 0;
    rounds[b] = 0;
    periodic[b] = 1;
}

// ------------------------------------------------------------------ driver
static void radix_sort(bz2hip_ctx *c, int nb, u32 maxn, int passes, u64 *&kin, u32 *&vin,
                       u64 *&kout, u32 *&vout) {
    Batch &B = c->batch;
    const u32 tiles = (maxn + RTILE - 1) / RTILE;
    dim3 grid(tiles, nb);
    for (int p = 0; p < passes; p++) {
        int shift = 8 * p;
        {
            ProfScope ps(c, P_RADIX_HIST);
            k2_radix_hist<<<grid, RT_THREADS, 0, c->stream>>>(kin, B.n, B.active, B.tilehist, B.S,
                                                              B.TPB, shift);
            KCHECK();
        }
        {
            ProfScope ps(c, P_RADIX_SCAN);
            k2_radix_scan<<<nb, 256, 0, c->stream>>>(B.n, B.active, B.tilehist, B.digitbase, B.TPB);
            KCHECK();
        }
        {
            ProfScope ps(c, P_RADIX_SCATTER);
            k2_radix_scatter<<<grid, RT_THREADS, 0, c->stream>>>(
                kin, vin, kout, vout, B.n, B.active, B.tilehist, B.digitbase, B.S, B.TPB, shift);
            KCHECK();
        }
        std::swap(kin, kout);
        std::swap(vin, vout);
    }
}

static void rerank(bz2hip_ctx *c, int nb, u32 maxn, const u64 *key, const u32 *val) {
    Batch &B = c->batch;
    const u32 tiles = (maxn + RTILE - 1) / RTILE;
    dim3 grid(tiles, nb);
    ProfScope ps(c, P_BWT_RERANK);
    k2_rerank_p1<<<grid, RT_THREADS, 0, c->stream>>>(key, B.n, B.active, B.tileagg, B.nheads, B.S,
                                                     B.TPB);
    KCHECK();
    k2_rerank_p2<<<nb, 256, 0, c->stream>>>(B.n, B.active, B.tileagg, B.TPB);
    KCHECK();
    k2_rerank_p3<<<grid, RT_THREADS, 0, c->stream>>>(key, val, B.n, B.active, B.tileagg, B.grp,
                                                     B.rank, B.sa, B.S, B.TPB);
    KCHECK();
}

// maxn = largest block length in the batch (host knows it)
void run_bwt_n(bz2hip_ctx *c, int nb, u32 maxn) {
    Batch &B = c->batch;
    if (nb == 0 || maxn == 0) return;
    const u32 tiles = (maxn + RTILE - 1) / RTILE;
    dim3 grid(tiles, nb);
    const int cb = (nb + 255) / 256;
    HIPCHK(hipMemsetAsync(B.ctl, 0, 4 * sizeof(u32), c->stream));
    k2_reset<<<cb, 256, 0, c->stream>>>(B.n, B.nheads, B.active, B.rounds, B.periodic, B.ctl, nb);
    KCHECK();
    {
        ProfScope ps(c, P_BWT_INIT);
        k2_init_keys<<<grid, RT_THREADS, 0, c->stream>>>(B.T, B.n, B.keyA, B.valA, B.S);
        KCHECK();
    }
    u64 *kin = B.keyA, *kout = B.keyB;
    u32 *vin = B.valA, *vout = B.valB;
    radix_sort(c, nb, maxn, 8, kin, vin, kout, vout);
    rerank(c, nb, maxn, kin, vin);
    u32 h = 8;
    for (;;) {
        HIPCHK(hipMemsetAsync(B.ctl, 0, 2 * sizeof(u32), c->stream));
        k2_round_ctl<<<cb, 256, 0, c->stream>>>(B.n, B.nheads, B.active, B.rounds, B.periodic, B.ctl,
                                                nb, h);
        KCHECK();
        u32 hc[3] = {0, 0, 0};
        HIPCHK(hipMemcpyAsync(hc, B.ctl, sizeof(hc), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (h == 8 && c->profile) { // round 0 sorted every element: 8 passes of 24 B
            c->prof_bytes[P_RADIX_SCATTER] += 8.0 * 24.0 * (double)hc[2];
            c->prof_bytes[P_RADIX_HIST] += 8.0 * 8.0 * (double)hc[2];
        }
        c->st_sorted_elems += (h == 8 ? (u64)hc[2] * 8 : 0) + (u64)hc[1] * (hc[0] ? 5 : 0);
        if (!hc[0]) break;
        if (c->profile) {
            c->prof_bytes[P_RADIX_SCATTER] += 5.0 * 24.0 * (double)hc[1];
            c->prof_bytes[P_RADIX_HIST] += 5.0 * 8.0 * (double)hc[1];
        }
        {
            ProfScope ps(c, P_BWT_KEYS);
            k2_build_keys<<<grid, RT_THREADS, 0, c->stream>>>(B.sa, B.grp, B.rank, B.n, B.active,
                                                              B.keyA, B.valA, B.S, h);
            KCHECK();
        }
        kin = B.keyA; kout = B.keyB; vin = B.valA; vout = B.valB;
        radix_sort(c, nb, maxn, 5, kin, vin, kout, vout);
        rerank(c, nb, maxn, kin, vin);
        h <<= 1;
    }
    {
        ProfScope ps(c, P_BWT_EMIT);
        k2_emit_bwt<<<grid, RT_THREADS, 0, c->stream>>>(B.T, B.sa, B.rank, B.n, B.periodic, B.bwt,
                                                        B.orig, B.S, c->opts.tie_rule);
        KCHECK();
    }
}
