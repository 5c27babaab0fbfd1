// k2_bwt.hip -- batched cyclic-rotation sort (BWT + origin pointer) by prefix doubling.
//
// Replaces BZip2DivSufSort.bwt() (S/BZip2DivSufSort.swift:3487-3506, sortTypeBstar :3076-3362,
// constructBWT :3369-3481).  The BWT string of a block is unique, so any correct rotation sort
// reproduces it; `orig` is unique unless the block is a perfect power w^k.
//
// Element word E (u64):  [59:40] grp = SA position of the element's group head
//                        [39:20] k2  = rank((i + h) mod n), filled in per round
//                        [19: 0] i   = rotation start
// Round 0: the first d0 symbols of every rotation (dense symbol codes as one number in base alphabet size, < 2^40)
//   are packed into bits [59:20]; 5 LSD radix passes of 8 bits over those 40 bits; ranks by head
//   flags.  Sorted depth h = d0.
// Round r: only rotations in groups of size > 1 are still listed.  Groups of <= GS elements that
//   lie inside one tile live in the S-list and are sorted by one workgroup in LDS (bitonic on E);
//   the others live in the L-list and take 5 global LSD passes.  After sorting, head flags split
//   the groups, ranks are refreshed, singletons write their BWT byte and leave, h doubles.
// A block ends when both lists are empty, or when h >= n (periodic block: the leftovers are the
// k-fold ties; they are flushed in list order).
// LSD pass = tile digit histogram (LDS atomics) -> per-block scan -> stable scatter (wave64
// ballot matching, per-wave LDS counters).  HBM-bound; no MFMA.
#include "ctx.hpp"

#include <chrono>
#include <climits>

#define GS 1024          // largest group the S-path takes
#define SC 3072          // S-tile nominal stride (window = SC + GS = 4096)
#define KEY_SHIFT 20
#define E_IMASK 0xFFFFFull
// long repeats at a fixed distance (see "long repeats" below; the round-0 re-rank uses the regions too)
#define VT 8192           // vote slots per block
#define NCAND 2           // distances kept per block
#define MAXRUNS 8         // regions kept per distance
#define MINRUN 4096u      // shortest gap of V worth recording (also > tile, so a tile holds no whole gap)
#define RUNS_STRIDE (4 + 4 * NCAND * MAXRUNS)
#define VSLOT(d) (((d) * 2654435761u) >> 19)

// XCD-aware launch: workgroups are dealt round-robin over the 8 XCDs by linear id, so block b
// is handled only by ids congruent to b mod 8: its rank array / lists stay in one XCD's L2.
#define XCD_MAP(gtiles, gnb)                                   \
    u32 bz_t, bz_b;                                            \
    if ((gnb) & 0x80000000u) {                                 \
        bz_t = blockIdx.x % (gtiles);                          \
        bz_b = blockIdx.x / (gtiles);                          \
    } else {                                                   \
        const u32 L_ = blockIdx.x, idx_ = L_ >> 3;             \
        bz_b = (L_ & 7u) + 8u * (idx_ / (gtiles));             \
        bz_t = idx_ % (gtiles);                                \
    }                                                          \
    if (bz_b >= ((gnb) & 0x7fffffffu)) return;
// plain mapping with the same interface (compute-bound kernels prefer even spread over XCDs)
static inline u32 xcd_grid(u32 tiles, u32 nb) { return 8u * tiles * ((nb + 7u) / 8u); }

enum { ST_DONE = 0, ST_ACTIVE = 1, ST_FIN = 2 };
enum { MODE_NORMAL = 0, MODE_ROUND0 = 1, MODE_FINAL = 2 };
enum { CLS_NONE = 0, CLS_SINGLE = 1, CLS_SMALL = 2, CLS_LKEEP = 3 };
#define CROSS INT_MAX
#define NOHEAD (INT_MIN / 2)

// Round-0 depth: the key is the first d0 symbols as ONE number in base `uniq` (the block's alphabet
// size), the largest d0 with uniq^d0 <= 2^r0bits: order-preserving like bit fields, but an alphabet
// of 17..101 symbols (text) gets 6 symbols into 40 bits instead of 5, one of 5 gets 17 instead of 13.
// What is left of the key space after d0 symbols takes a coarse, order-preserving bucket of the next
// symbol: B = min(uniq, floor(2^r0bits / uniq^d0)) buckets, bucket(code) = code * scale >> 16 with
// scale = (B << 16) / uniq (text, 55 symbols: d0 = 6 and 39 buckets of the seventh).  The depth that
// the doubling counts on stays d0.  *xb = B | scale << 8 (B <= 255 here: uniq = 256 leaves B = 1).
__device__ __forceinline__ u32 r0_depth(u32 uniq, u32 r0bits, u32 *xb = nullptr) {
    if (xb) *xb = 1u;
    if (uniq <= 2) return r0bits;
    const u64 lim = 1ull << r0bits;
    u64 p = uniq;
    u32 d = 1;
    while (p * uniq <= lim && d < 63) { p *= uniq; d++; } // (keys are < uniq^d0 <= 2^r0bits)
    if (xb) {
        const u32 B = (u32)min((u64)min(uniq, 255u), lim / p);
        *xb = B | (((B << 16) / uniq) << 8);
    }
    return d;
}

// ------------------------------------------------------------------ per-block control
__global__ void k2_reset(const u32 *__restrict__ nArr, const u32 *__restrict__ present,
                         u32 *mS, u32 *mL, u32 *mSn, u32 *mLn, u32 *hcur, u32 *huse, u32 *state,
                         u32 *par, u32 *rounds, u32 *leftg, u32 *periodic, u32 *gcount, u32 *ctl, u32 nb,
                         u32 r0bits) {
    u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    u32 uniq = 0;
    for (int q = 0; q < 8; q++) uniq += __popc(present[b * 8 + q]);
    u32 d0 = r0_depth(uniq, r0bits);
    mS[b] = 0; mL[b] = 0; mSn[b] = 0; mLn[b] = 0;
    hcur[b] = d0; huse[b] = d0;
    state[b] = nArr[b] > 0 ? ST_ACTIVE : ST_DONE;
    par[b] = 0; rounds[b] = 0; leftg[b] = 0; periodic[b] = 1; gcount[b] = 0;
    atomicAdd(&ctl[8], nArr[b]);
}

// after every round: move the "next" counters in, decide who continues.  One workgroup per block.
// A block whose rotations are ALL still listed, in g groups of equal size k = n / g >= 2, may be a
// perfect power w^k with |w| = g: if T has the cyclic period g, rotations i and i + g are the same
// string, every group holds at least one such class, and g groups over n rotations leave room for
// exactly one class each -- the groups are the k-fold ties and can never split.  The block is then
// finished now (its ties are flushed in list order) instead of doubling on to h >= n: "ab...",
// zeros and the like take one or two rounds instead of log2(n).
// Control words.  ctl[0..16) live for the whole sort: [4] any block finished as periodic, [5] / [6]
// largest S- / L-list among those, [8] sum of the block lengths, [13..16) the tie replay's checks.
// The words of ONE round sit in ctl[16 + 16 * (round & 1) ..): [0] any block goes on, [1] / [2] largest
// S- / L-list, [3] / [4] elements in all L- / S-lists, [5] largest S + L of a block (bounds the next
// round's lists: a block's lists only shrink, and nothing enters an L-list from an S-list).  A round
// zeroes the other parity's words for its successor, so the loop needs no memset and the host can
// read a round's words (copied in stream order) while later rounds are already queued.
#define RC_ANY 0
#define RC_MAXS 1
#define RC_MAXL 2
#define RC_SUML 3
#define RC_SUMS 4
#define RC_MAXSL 5
#define CTL_WORDS 48
__global__ __launch_bounds__(256) void k2_ctl(const u32 *__restrict__ nArr, u32 *mS, u32 *mL, u32 *mSn, u32 *mLn,
                                              u32 *hcur, u32 *huse, u32 *state, u32 *par, u32 *rounds,
                                              u32 *gcount, const u8 *__restrict__ T, u32 S, u32 *ctl, u32 *rc,
                                              u32 *rc_next, u32 nb) {
    const u32 b = blockIdx.x;
    if (b >= nb) return;
    if (b == 0 && threadIdx.x < 8) rc_next[threadIdx.x] = 0;
    if (state[b] != ST_ACTIVE) return;
    __shared__ u32 sh[6];
    __shared__ u32 s_bad;
    if (threadIdx.x == 0) {
        sh[0] = mSn[b]; sh[1] = mLn[b]; sh[2] = nArr[b]; sh[3] = hcur[b]; sh[4] = gcount[b];
        s_bad = 0;
    }
    __syncthreads();
    const u32 s = sh[0], l = sh[1], n = sh[2], h = sh[3], g = sh[4];
    bool fin = (s + l != 0) && h >= n;
    if (!fin && s + l == n && g >= 1 && n % g == 0 && n / g >= 2) {
        const u8 *t = T + (size_t)b * S;
        u32 bad = 0;
        for (u32 i = threadIdx.x; i < n && !bad; i += 256) {
            u32 j = i + g;
            if (j >= n) j -= n;
            bad |= (t[i] != t[j]);
        }
        if (bad) atomicOr(&s_bad, 1u);
        __syncthreads();
        fin = s_bad == 0;
    }
    if (threadIdx.x != 0) return;
    mS[b] = s; mL[b] = l; mSn[b] = 0; mLn[b] = 0;
    gcount[b] = 0;
    par[b] ^= 1u; // survivors were written to the other S buffer
    if (s + l == 0) {
        state[b] = ST_DONE;
    } else if (fin) {
        state[b] = ST_FIN;
        atomicOr(&ctl[4], 1u);
        atomicMax(&ctl[5], s);
        atomicMax(&ctl[6], l);
    } else {
        huse[b] = h;
        hcur[b] = (h > (1u << 30)) ? h : 2 * h;
        rounds[b] += 1;
        atomicOr(&rc[RC_ANY], 1u);
        atomicMax(&rc[RC_MAXS], s);
        atomicMax(&rc[RC_MAXL], l);
        atomicAdd(&rc[RC_SUML], l);
        atomicAdd(&rc[RC_SUMS], s);
        atomicMax(&rc[RC_MAXSL], s + l);
    }
}


// Round-0 keys straight from the block bytes (first radix pass only): dense symbol map from the
// present bitmap, the tile's bytes (+ the d0 - 1 that follow, cyclically) staged in LDS.
struct R0Tile {
    u8 smap[256];
    __attribute__((aligned(16))) u8 t[RTILE + 64]; // the tile's bytes as DENSE symbols (translated once, at the load)
    u32 uniq;
};
__device__ __forceinline__ void r0_tile_load(R0Tile &L, const u8 *__restrict__ T,
                                             const u32 *__restrict__ present, u32 b, u32 S, u32 n,
                                             u32 base) {
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    {
        u32 v = threadIdx.x;
        const u32 *pw = present + b * 8;
        u32 cnt = 0;
        for (u32 wi = 0; wi < (v >> 5); wi++) cnt += __popc(pw[wi]);
        cnt += __popc(pw[v >> 5] & ((1u << (v & 31)) - 1u));
        L.smap[v] = (u8)cnt;
        if (v == 255) L.uniq = cnt + ((pw[7] >> 31) & 1u);
    }
    __syncthreads();
    // 16 bytes per thread and trip: one vector load where the 16 lie below n (base is a multiple of the
    // tile and the slot is 16-byte aligned), single bytes with the cyclic continuation at the block's end
    const u8 *t = T + (size_t)b * S;
    for (u32 u = threadIdx.x; u < (RTILE + 64) / 16; u += RT_THREADS) {
        const u32 p0 = base + u * 16u;
        u32 wv[4];
        if (p0 + 16u <= n) {
            const u32x4 v = *(const u32x4 *)(t + p0);
            wv[0] = v.x; wv[1] = v.y; wv[2] = v.z; wv[3] = v.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                u32 x = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    u32 p = p0 + q * 4 + j;
                    if (p >= n) p = (p - n) % n; // (n can be smaller than the overhang)
                    x |= (u32)t[p] << (8 * j);
                }
                wv[q] = x;
            }
        }
        u32 ov[4];
#pragma unroll
        for (int q = 0; q < 4; q++)
            ov[q] = (u32)L.smap[wv[q] & 255u] | ((u32)L.smap[(wv[q] >> 8) & 255u] << 8) |
                    ((u32)L.smap[(wv[q] >> 16) & 255u] << 16) | ((u32)L.smap[wv[q] >> 24] << 24);
        *(u32x4 *)&L.t[u * 16u] = (u32x4){ov[0], ov[1], ov[2], ov[3]};
    }
    __syncthreads();
}
// The key of rotation i (tile-local position `local`): its first d0 symbols as one number in base uniq,
// then the bucket of the next symbol.  Alphabets of 33 symbols and more have d0 <= 7: the eight
// symbols from `local` on come out of three LDS words, the first four fold in 32 bits.  Smaller
// alphabets (up to 40 symbols deep) take the loop.
__device__ __forceinline__ u64 r0_key(const R0Tile &L, u32 local, u32 i, u32 d0, u32 xb) {
    u64 kk;
    const u32 uniq = L.uniq;
    u32 nxt; // the symbol behind the d0-th
    if (d0 >= 4u && d0 <= 7u) {
        const u32 *t32 = (const u32 *)L.t;
        const u32 w = local >> 2, a = local & 3u;
        const u32 w0 = t32[w], w1 = t32[w + 1], w2 = t32[w + 2];
        const u32 lo = __builtin_amdgcn_alignbyte(w1, w0, a), hi = __builtin_amdgcn_alignbyte(w2, w1, a);
        // first four symbols in 32 bits, the rest and the bucket of the next symbol in 24 (every step a
        // full-rate 24-bit multiply-add: the operands stay below 2^24), one 64-bit multiply-add joins them
        u32 k32 = lo & 255u;
        k32 = __umul24(k32, uniq) + ((lo >> 8) & 255u);
        k32 = __umul24(k32, uniq) + ((lo >> 16) & 255u);
        k32 = __umul24(k32, uniq) + (lo >> 24); // < uniq^4 <= 2^32 (the product's low 32 bits are exact)
        u32 h = hi, klo = 0, M = 1; // M = uniq^(d0 - 4) * B <= 2^40 / uniq^4 <= 2^20 (uniq >= 33 on this path)
        for (u32 q = 4; q < d0; q++) {
            klo = __umul24(klo, uniq) + (h & 255u);
            M *= uniq;
            h >>= 8;
        }
        nxt = h & 255u;
        const u32 Bq = xb & 255u;
        if (Bq > 1u) {
            klo = __umul24(klo, Bq) + ((nxt * (xb >> 8)) >> 16);
            M *= Bq;
        }
        return (((u64)k32 * M + klo) << KEY_SHIFT) | (u64)i;
    } else {
        kk = 0;
        for (u32 q = 0; q < d0; q++) kk = kk * uniq + L.t[local + q];
        nxt = L.t[local + d0];
    }
    const u32 B = xb & 255u;
    if (B > 1u) kk = kk * B + ((nxt * (xb >> 8)) >> 16);
    return (kk << KEY_SHIFT) | (u64)i;
}

// The key's low 24 bits alone (the first pass's histogram needs only its lowest digit): the same
// sum of products modulo 2^24, every step one 24-bit multiply-add instead of 64-bit arithmetic.
__device__ __forceinline__ u32 r0_key_low24(const R0Tile &L, u32 local, u32 d0, u32 xb) {
    const u32 uniq = L.uniq;
    u32 k, nxt;
    if (d0 >= 4u && d0 <= 7u) {
        const u32 *t32 = (const u32 *)L.t;
        const u32 w = local >> 2, a = local & 3u;
        const u32 w0 = t32[w], w1 = t32[w + 1], w2 = t32[w + 2];
        const u32 lo = __builtin_amdgcn_alignbyte(w1, w0, a), hi = __builtin_amdgcn_alignbyte(w2, w1, a);
        k = lo & 255u;
        k = __umul24(k, uniq) + ((lo >> 8) & 255u);
        k = __umul24(k, uniq) + ((lo >> 16) & 255u);
        k = __umul24(k, uniq) + (lo >> 24);
        u32 h = hi;
        for (u32 q = 4; q < d0; q++) {
            k = __umul24(k, uniq) + (h & 255u);
            h >>= 8;
        }
        nxt = h & 255u;
    } else {
        k = 0;
        for (u32 q = 0; q < d0; q++) k = __umul24(k, uniq) + L.t[local + q];
        nxt = L.t[local + d0];
    }
    const u32 B = xb & 255u;
    if (B > 1u) k = __umul24(k, B) + ((nxt * (xb >> 8)) >> 16);
    return k;
}

// ------------------------------------------------------------------ LSD radix pass (u64 words)
// DB = digit bits (8 or 10), RB = bins.  Per pass: tile digit histogram -> per-block scan -> stable
// scatter.  The scatter first orders the tile by digit in LDS (rank = bin start + wave offset +
// rank inside the wave's bin, from wave64 ballots) and then stores it in that order: consecutive
// lanes write consecutive addresses of one bin, so a wave store touches a handful of lines instead
// of one line per lane (the stores were bound by L2 requests, not by bytes).
template <bool FROM_T, int DB>
__global__ __launch_bounds__(RT_THREADS) void k2_radix_hist(const u64 *__restrict__ key,
                                                            const u32 *__restrict__ cntArr,
                                                            const u32 *__restrict__ state,
                                                            u32 *__restrict__ tilehist, u32 S,
                                                            u32 TPB, int shift, u32 gtiles, u32 gnb,
                                                            const u8 *__restrict__ T,
                                                            const u32 *__restrict__ present, u32 r0bits,
                                                            const u8 *__restrict__ dig) {
    constexpr u32 RB = 1u << DB;
    XCD_MAP(gtiles, gnb)
    const u32 b = bz_b;
    if (state[b] != ST_ACTIVE) return;
    const u32 n = cntArr[b];
    const u32 base = bz_t * RTILE;
    if (base >= n) return;
    // Lanes of one LDS atomic that hit one word are served one after the other, and digits are
    // anything but uniform (text, periodic data, zeros: 91 % of this kernel's LDS cycles were such
    // conflicts): every bin exists HR times, lane l counts in copy l mod HR, and a thread that meets
    // the same digit several times in a row adds the run at once.
    constexpr u32 HR = DB == 8 ? 8u : 1u;
    __shared__ u32 hist[RB * HR];
    for (u32 q = threadIdx.x; q < RB * HR; q += RT_THREADS) hist[q] = 0;
    u32 *th = tilehist + ((size_t)b * TPB + bz_t) * RB;
    const u32 hr = threadIdx.x & (HR - 1u);
    auto flush = [&]() {
        __syncthreads();
        for (u32 q = threadIdx.x; q < RB; q += RT_THREADS) {
            u32 sum = 0;
#pragma unroll
            for (u32 r = 0; r < HR; r++) sum += hist[q * HR + r];
            th[q] = sum;
        }
    };
    if (FROM_T) {
        __shared__ R0Tile L;
        r0_tile_load(L, T, present, b, S, n, base);
        u32 xb;
        const u32 d0 = r0_depth(L.uniq, r0bits, &xb);
#pragma unroll 4
        for (int k = 0; k < RT_ITEMS; k++) {
            const u32 local = k * RT_THREADS + threadIdx.x;
            if (base + local < n) // (this kernel builds keys only for the first pass: shift == KEY_SHIFT, DB <= 24)
                atomicAdd(&hist[(r0_key_low24(L, local, d0, xb) & (RB - 1u)) * HR + hr], 1u);
        }
        flush();
        return;
    }
    __syncthreads();
    if (!FROM_T && DB == 8 && dig) {
        // the scatter of the pass before left this pass's digit of every element, in list order, as one
        // byte each: 1 B per element to read instead of the 8-byte word
        typedef u32 u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_nontemporal_load((const u32x4 *)(dig + (size_t)b * S + base) + threadIdx.x);
        const u32 wv[4] = {v.x, v.y, v.z, v.w};
        const u32 i0 = base + 16u * threadIdx.x;
        u32 prev = wv[0] & 255u, run = i0 < n ? 1u : 0u;
#pragma unroll
        for (int q = 1; q < 16; q++) {
            const u32 d = (wv[q >> 2] >> (8 * (q & 3))) & 255u;
            const bool in = i0 + q < n;
            if (in && d != prev) {
                atomicAdd(&hist[prev * HR + hr], run);
                prev = d;
                run = 0;
            }
            run += in ? 1u : 0u;
        }
        if (run) atomicAdd(&hist[prev * HR + hr], run);
        flush();
        return;
    }
    const u64 *kp = key + (size_t)b * S;
    typedef u64 u64x2 __attribute__((ext_vector_type(2)));
    const u64x2 *kp2 = (const u64x2 *)(kp + base); // base is a multiple of 4096: 16-byte aligned
    // the block slot is padded (S >= n + 8 and a multiple of the tile), so whole pairs can be
    // read unconditionally; all 8 loads are issued before the first LDS atomic
    u64x2 vv[RT_ITEMS / 2];
#pragma unroll
    for (int k = 0; k < RT_ITEMS / 2; k++) vv[k] = __builtin_nontemporal_load(&kp2[k * RT_THREADS + threadIdx.x]);
#pragma unroll
    for (int k = 0; k < RT_ITEMS / 2; k++) {
        const u32 i = base + 2 * (k * RT_THREADS + threadIdx.x);
        if (i < n) atomicAdd(&hist[((u32)(vv[k].x >> shift) & (RB - 1u)) * HR + hr], 1u);
        if (i + 1 < n) atomicAdd(&hist[((u32)(vv[k].y >> shift) & (RB - 1u)) * HR + hr], 1u);
    }
    flush();
}

template <int DB>
__global__ __launch_bounds__(256) void k2_radix_scan(const u32 *__restrict__ cntArr,
                                                     const u32 *__restrict__ state,
                                                     u32 *__restrict__ tilehist,
                                                     u32 *__restrict__ digitbase, u32 TPB) {
    constexpr u32 RB = 1u << DB;
    constexpr int PER = RB / 256; // digits per thread (consecutive)
    const u32 b = blockIdx.x;
    if (state[b] != ST_ACTIVE) return;
    const u32 n = cntArr[b];
    const u32 tiles = (n + RTILE - 1) / RTILE;
    __shared__ u32 sm[8];
    u32 run[PER];
#pragma unroll
    for (int q = 0; q < PER; q++) run[q] = 0;
    // 8 tiles at a time: the loads of a chunk are issued together (the walk is a dependent chain
    // only through the running sums)
    constexpr int TB = 8;
    for (u32 t0 = 0; t0 < tiles; t0 += TB) {
        u32 cv[TB][PER];
#pragma unroll
        for (int k = 0; k < TB; k++)
#pragma unroll
            for (int q = 0; q < PER; q++) {
                const u32 t = min(t0 + (u32)k, tiles - 1u);
                // digit = q * 256 + thread: coalesced over threads
                cv[k][q] = tilehist[((size_t)b * TPB + t) * RB + q * 256 + threadIdx.x];
            }
#pragma unroll
        for (int k = 0; k < TB; k++) {
            if (t0 + (u32)k >= tiles) break;
#pragma unroll
            for (int q = 0; q < PER; q++) {
                tilehist[((size_t)b * TPB + t0 + k) * RB + q * 256 + threadIdx.x] = run[q];
                run[q] += cv[k][q];
            }
        }
    }
    // exclusive scan over digits in digit order: digit = q * 256 + thread
    u32 carry = 0;
#pragma unroll
    for (int q = 0; q < PER; q++) {
        u32 total;
        u32 ex = block_excl_scan_add<256>(run[q], sm, &total);
        digitbase[b * RB + q * 256 + threadIdx.x] = carry + ex;
        carry += total;
        __syncthreads();
    }
}

// Lanes holding the same DB-bit digit (stable ranking inside a wave).  Per bit: t = -bit (one
// v_bfe_i32), one ballot, and "lanes that differ from me in this bit" = ballot ^ t, OR-ed up;
// the complement of the union is the match set.
template <int DB>
__device__ __forceinline__ u64 match_digit(u32 d, bool valid) {
    u32 acc_lo = 0, acc_hi = 0;
#pragma unroll
    for (int bit = 0; bit < DB; bit++) {
        const int t = __builtin_amdgcn_sbfe((int)d, bit, 1); // 0 or -1
        const u64 bb = __ballot(t != 0);
        acc_lo |= (u32)bb ^ (u32)t;
        acc_hi |= (u32)(bb >> 32) ^ (u32)t;
    }
    const u64 v = __ballot(valid);
    return ~(((u64)acc_hi << 32) | (u64)acc_lo) & v;
}

template <bool FROM_T, int DB, bool STAGE>
__global__ __launch_bounds__(RT_THREADS, 4) void k2_radix_scatter(
    const u64 *__restrict__ keyIn, u64 *__restrict__ keyOut, const u32 *__restrict__ cntArr,
    const u32 *__restrict__ state, const u32 *__restrict__ tilehist,
    const u32 *__restrict__ digitbase, u32 S, u32 TPB, int shift, u32 gtiles, u32 gnb,
    const u8 *__restrict__ T, const u32 *__restrict__ present, u32 r0bits, u8 *__restrict__ digOut) {
    constexpr u32 RB = 1u << DB;
    constexpr int PER = RB / 256;
    XCD_MAP(gtiles, gnb)
    const u32 b = bz_b;
    if (state[b] != ST_ACTIVE) return;
    const u32 n = cntArr[b];
    const u32 base = bz_t * RTILE;
    if (base >= n) return;
    __shared__ u32 whist[4][RB];   // per-wave bin counts -> exclusive offsets over waves
    __shared__ u32 binstart[STAGE ? RB : 1];   // tile-local start of every bin
    __shared__ u32 gofs[RB];                   // output slot of the tile's first element of every bin
    __shared__ u64 stage[STAGE ? RTILE : 1];   // the tile ordered by digit
    __shared__ u32 sm[8];
    const int w = wave_id(), lane = lane_id();
    for (int q = threadIdx.x; q < 4 * (int)RB; q += RT_THREADS) (&whist[0][0])[q] = 0;
    __shared__ R0Tile L;
    u32 d0 = 0, xb = 1;
    if (FROM_T) {
        r0_tile_load(L, T, present, b, S, n, base);
        d0 = r0_depth(L.uniq, r0bits, &xb);
    }
    __syncthreads();
    const u64 *kp = keyIn + (size_t)b * S;
    u64 keys[RT_ITEMS];
    u32 lr[RT_ITEMS];
    const u64 lt = (1ull << lane) - 1ull;
    // all 16 loads first (clamped address instead of a branch, so that they are issued back to
    // back and the ranking below does not wait for memory once per row)
#pragma unroll
    for (int k = 0; k < RT_ITEMS; k++) {
        const u32 i = base + w * (RTILE / 4) + k * 64 + lane;
        const u32 ic = i < n ? i : base; // base < n
        keys[k] = FROM_T ? r0_key(L, ic - base, ic, d0, xb) : kp[ic];
    }
    // Ranking, in two loops so that nothing waits inside them: the leaders' LDS atomics (with
    // return) of all 16 rows are issued back to back -- the LDS executes one wave's atomics in
    // order, so the counts stay stable -- and the results are broadcast afterwards.
    u32 oldv[RT_ITEMS];
#pragma unroll
    for (int k = 0; k < RT_ITEMS; k++) {
        u32 i = base + w * (RTILE / 4) + k * 64 + lane;
        bool valid = i < n;
        u64 kk = valid ? keys[k] : 0;
        keys[k] = kk;
        u32 d = (u32)(kk >> shift) & (RB - 1u);
        u64 m = match_digit<DB>(d, valid);
        const int leader = __ffsll((unsigned long long)m) - 1;
        lr[k] = __popcll(m & lt) | ((u32)(leader < 0 ? 0 : leader) << 8); // rank in the match set | leader lane
        oldv[k] = 0;
        if (valid && lane == leader) oldv[k] = atomicAdd(&whist[w][d], (u32)__popcll(m));
    }
#pragma unroll
    for (int k = 0; k < RT_ITEMS; k++) lr[k] = (lr[k] & 255u) + __shfl(oldv[k], (int)(lr[k] >> 8), 64);
    __syncthreads();
    {
        // per digit: exclusive offsets over the 4 waves, the tile's bin sizes -> bin starts
        u32 tot[PER];
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const u32 d = threadIdx.x * PER + q; // consecutive digits per thread: scan order
            u32 run = 0;
#pragma unroll
            for (int ww = 0; ww < 4; ww++) {
                u32 c = whist[ww][d];
                whist[ww][d] = run;
                run += c;
            }
            tot[q] = run;
            gofs[d] = digitbase[b * RB + d] + tilehist[((size_t)b * TPB + bz_t) * RB + d];
        }
        if (STAGE) {
            u32 mine = 0;
#pragma unroll
            for (int q = 0; q < PER; q++) mine += tot[q];
            u32 total;
            u32 ex = block_excl_scan_add<256>(mine, sm, &total);
#pragma unroll
            for (int q = 0; q < PER; q++) {
                binstart[threadIdx.x * PER + q] = ex;
                ex += tot[q];
            }
        }
    }
    __syncthreads();
    u64 *ko = keyOut + (size_t)b * S;
    // the next pass's digit of every element, at the element's new slot (one byte each: that pass's
    // histogram then reads 1 B per element instead of 8)
    u8 *dg = digOut ? digOut + (size_t)b * S : nullptr;
    if (!STAGE) {
        // straight from the registers: a bin's elements of one wave row go to consecutive slots
#pragma unroll
        for (int k = 0; k < RT_ITEMS; k++) {
            u32 i = base + w * (RTILE / 4) + k * 64 + lane;
            if (i < n) {
                u32 d = (u32)(keys[k] >> shift) & (RB - 1u);
                ko[gofs[d] + whist[w][d] + lr[k]] = keys[k]; // (not nontemporal: L2 merges these into full lines)
                if (dg) dg[gofs[d] + whist[w][d] + lr[k]] = (u8)(keys[k] >> (shift + DB));
            }
        }
        return;
    }
    // the tile in digit order
#pragma unroll
    for (int k = 0; k < RT_ITEMS; k++) {
        u32 i = base + w * (RTILE / 4) + k * 64 + lane;
        if (i < n) {
            u32 d = (u32)(keys[k] >> shift) & (RB - 1u);
            stage[binstart[d] + whist[w][d] + lr[k]] = keys[k];
        }
    }
    __syncthreads();
    const u32 cnt = min((u32)RTILE, n - base);
#pragma unroll
    for (int k = 0; k < RT_ITEMS; k++) {
        const u32 j = k * RT_THREADS + threadIdx.x;
        if (j < cnt) {
            const u64 kk = stage[j];
            const u32 d = (u32)(kk >> shift) & (RB - 1u);
            ko[gofs[d] + (j - binstart[d])] = kk; // (not nontemporal: L2 completes the lines shared by two tiles)
            if (dg) dg[gofs[d] + (j - binstart[d])] = (u8)(kk >> (shift + DB));
        }
    }
}

// ------------------------------------------------------------------ doubling keys
// z = 0: S-list, z = 1: L-list.  E |= rank((i + h) mod n) << 20
__global__ __launch_bounds__(RT_THREADS) void k2_keys(u64 *__restrict__ slA, u64 *__restrict__ slB,
                                                      u64 *__restrict__ keyA,
                                                      const u32 *__restrict__ par,
                                                      const u32 *__restrict__ mS,
                                                      const u32 *__restrict__ mL,
                                                      const u32 *__restrict__ state,
                                                      const u32 *__restrict__ huse,
                                                      const u32 *__restrict__ nArr,
                                                      const u32 *__restrict__ rank, u32 S, u32 gtiles, u32 gnb) {
    XCD_MAP(2 * gtiles, gnb)
    const u32 b = bz_b;
    const u32 zsel = bz_t / gtiles;
    bz_t = bz_t % gtiles;
    if (state[b] != ST_ACTIVE) return;
    const u32 m = zsel ? mL[b] : mS[b];
    const u32 base = bz_t * RTILE;
    if (base >= m) return;
    u64 *list = (zsel ? keyA : (par[b] ? slB : slA)) + (size_t)b * S;
    const u32 n = nArr[b], h = huse[b];
    const u32 *rk = rank + (size_t)b * S;
    // unconditional phases (clamped addresses) so that 8 loads, then 8 gathers, are in flight
    // together instead of one dependent load-gather-store chain per element; two halves keep the
    // register count low enough for 8 waves per SIMD
#pragma unroll
    for (int half = 0; half < 2; half++) {
        u64 ev[RT_ITEMS / 2];
        u32 rv[RT_ITEMS / 2];
#pragma unroll
        for (int k = 0; k < RT_ITEMS / 2; k++) {
            const u32 j = base + (half * (RT_ITEMS / 2) + k) * RT_THREADS + threadIdx.x;
            ev[k] = list[j < m ? j : base];
        }
#pragma unroll
        for (int k = 0; k < RT_ITEMS / 2; k++) {
            u32 i2 = (u32)(ev[k] & E_IMASK) + h;
            if (i2 >= n) i2 -= n;
            rv[k] = rk[i2];
        }
#pragma unroll
        for (int k = 0; k < RT_ITEMS / 2; k++) {
            const u32 j = base + (half * (RT_ITEMS / 2) + k) * RT_THREADS + threadIdx.x;
            if (j < m) list[j] = ev[k] | ((u64)rv[k] << KEY_SHIFT);
        }
    }
}

// ------------------------------------------------------------------ tile classification
// buf[0..cnt) holds a run of the (sorted) list in LDS; wave-row layout: wave w owns
// [w*1024, w*1024+1024), 16 rows of 64.  Computes for every element the slot of its old / new
// group head (tile-local, may be negative = before the tile) and of the next new head.
// LDS index swizzle of the S-sort window: lets a thread own 16 consecutive elements (register
// stages of the bitonic network) without every lane hitting the same banks
__device__ __forceinline__ int SZ(int i) { return i ^ ((i >> 4) & 15); }
#define BUFX(i) buf[SWZ ? SZ(i) : (i)]

template <int MODE, bool SWZ = false>
__device__ __forceinline__ void tile_classify(const u64 *buf, int cnt, u64 prevE, bool has_prev,
                                              int carryOld, int carryNew, bool end_head,
                                              int *sm /* 12 ints */, int (&cls)[16],
                                              u32 (&ngrp)[16], u32 (&spos)[16], int &nheads,
                                              int *gsz = nullptr /* [16]: size of the element's group if it lies inside the tile, else 0 */) {
    const int w = wave_id(), lane = lane_id();
    u64 nm[16], om[16];
    int lastOld = NOHEAD, lastNew = NOHEAD, firstNew = CROSS;
    int heads = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int rb = w * 1024 + k * 64;
        const int idx = rb + lane;
        bool nh = false, oh = false;
        if (idx < cnt) {
            u64 e = BUFX(idx);
            if (idx == 0) {
                if (!has_prev) {
                    nh = true;
                    oh = true;
                } else {
                    nh = (e >> KEY_SHIFT) != (prevE >> KEY_SHIFT);
                    oh = (MODE == MODE_ROUND0) ? false : ((e >> 40) != (prevE >> 40));
                }
            } else {
                u64 ep = BUFX(idx - 1);
                nh = (e >> KEY_SHIFT) != (ep >> KEY_SHIFT);
                oh = (MODE == MODE_ROUND0) ? false : ((e >> 40) != (ep >> 40));
            }
        }
        u64 a = __ballot(nh), o = __ballot(oh);
        nm[k] = a;
        om[k] = o;
        if (o) lastOld = rb + 63 - __clzll((unsigned long long)o);
        if (a) {
            lastNew = rb + 63 - __clzll((unsigned long long)a);
            if (firstNew == CROSS) firstNew = rb + __ffsll((unsigned long long)a) - 1;
            heads += __popcll(a);
        }
    }
    __syncthreads();
    if (lane == 0) {
        sm[w] = lastOld;
        sm[4 + w] = lastNew;
        sm[8 + w] = firstNew;
        sm[12 + w] = heads;
    }
    __syncthreads();
    int cO = carryOld, cN = carryNew;
    for (int q = 0; q < w; q++) {
        if (sm[q] != NOHEAD) cO = sm[q];
        if (sm[4 + q] != NOHEAD) cN = sm[4 + q];
    }
    int nx = end_head ? cnt : CROSS;
    for (int q = 3; q > w; q--)
        if (sm[8 + q] != CROSS) nx = sm[8 + q];
    nheads = sm[12] + sm[13] + sm[14] + sm[15];
    int ohi[16], nhi[16];
    const u64 le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int rb = w * 1024 + k * 64;
        u64 mo = om[k] & le, mn = nm[k] & le;
        ohi[k] = mo ? rb + 63 - __clzll((unsigned long long)mo) : cO;
        nhi[k] = mn ? rb + 63 - __clzll((unsigned long long)mn) : cN;
        if (om[k]) cO = rb + 63 - __clzll((unsigned long long)om[k]);
        if (nm[k]) cN = rb + 63 - __clzll((unsigned long long)nm[k]);
    }
    const u64 gt = (lane == 63) ? 0ull : (~0ull << (lane + 1));
#pragma unroll
    for (int k = 15; k >= 0; k--) {
        const int rb = w * 1024 + k * 64;
        const int idx = rb + lane;
        u64 mn = nm[k] & gt;
        int next = mn ? rb + __ffsll((unsigned long long)mn) - 1 : nx;
        if (nm[k]) nx = rb + __ffsll((unsigned long long)nm[k]) - 1;
        int c = CLS_NONE;
        u32 g = 0, sp = 0;
        if (idx < cnt) {
            u64 e = BUFX(idx);
            u32 gb = (MODE == MODE_ROUND0) ? 0u : (u32)(e >> 40);
            g = gb + (u32)(nhi[k] - ohi[k]);
            sp = gb + (u32)(idx - ohi[k]);
            if (nhi[k] < 0 || next == CROSS) {
                c = CLS_LKEEP;
                if (gsz) gsz[k] = 0;
            } else {
                int size = next - nhi[k];
                c = (size == 1) ? CLS_SINGLE : (size <= GS ? CLS_SMALL : CLS_LKEEP);
                if (gsz) gsz[k] = size;
            }
        } else if (gsz) {
            gsz[k] = 0;
        }
        cls[k] = c;
        ngrp[k] = g;
        spos[k] = sp;
    }
}

// shared output stage: ranks, BWT bytes of finished rotations, survivors to the S / L lists
template <int MODE, bool SWZ = false>
__device__ __forceinline__ void tile_emit(const u64 *buf, int cnt, const int (&cls)[16],
                                          const u32 (&ngrp)[16], const u32 (&spos)[16],
                                          u32 *__restrict__ rank, const u8 *__restrict__ T,
                                          u8 *__restrict__ bwt, u32 n, u64 *__restrict__ sOut,
                                          u32 *mSn_b, u64 *__restrict__ lOut, u32 lBase,
                                          int *sm /* 16 ints */, u32 *gcount_b, u32 rlo = 0, u32 rhi = ~0u,
                                          bool mainSweep = true) {
    const int w = wave_id(), lane = lane_id();
    if (!mainSweep) {
        // a later sweep of round 0: only the ranks of positions in [rlo, rhi)
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int idx = w * 1024 + k * 64 + lane;
            if (cls[k] != CLS_NONE) {
                const u32 i = (u32)(BUFX(idx) & E_IMASK);
                if (i >= rlo && i < rhi) rank[i] = ngrp[k];
            }
        }
        return;
    }
    // counts of SMALL and LKEEP per wave, and of the groups that stay listed (a survivor whose
    // slot is its group's rank is that group's head)
    u32 cs = 0, cl = 0, ch = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        cs += (cls[k] == CLS_SMALL);
        cl += (cls[k] == CLS_LKEEP);
        ch += ((cls[k] == CLS_SMALL || cls[k] == CLS_LKEEP) && spos[k] == ngrp[k]);
    }
    if (MODE != MODE_FINAL) {
        cs = wave_reduce_add(cs);
        cl = wave_reduce_add(cl);
        ch = wave_reduce_add(ch);
        __syncthreads();
        if (lane == 0) {
            sm[w] = (int)cs;
            sm[4 + w] = (int)cl;
            sm[12 + w] = (int)ch;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            u32 tot = (u32)(sm[0] + sm[1] + sm[2] + sm[3]);
            sm[8] = tot ? (int)atomicAdd(mSn_b, tot) : 0;
            u32 th = (u32)(sm[12] + sm[13] + sm[14] + sm[15]);
            if (th) atomicAdd(gcount_b, th);
        }
        __syncthreads();
    }
    u32 sRun = 0, lRun = 0;
    if (MODE != MODE_FINAL) {
        sRun = (u32)sm[8];
        lRun = lBase;
        for (int q = 0; q < w; q++) {
            sRun += (u32)sm[q];
            lRun += (u32)sm[4 + q];
        }
    }
    // The BWT bytes of rotations that finish in a doubling round are NOT written here: a finished
    // rotation's rank is its final sorted position, so one streaming pass at the end of the sort
    // (k2_emit_bwt: bwt[rank[i]] = T[i - 1]) writes them all, instead of one random T gather per
    // rotation here.  Only the k-fold ties of a periodic block (MODE_FINAL) are placed from the
    // list, in list order: gather their preceding bytes first (16 in flight), then store.
    u8 tb[16];
    if (MODE == MODE_FINAL) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int idx = w * 1024 + k * 64 + lane;
            u32 a = 0;
            if (cls[k] != CLS_NONE) {
                const u32 i = (u32)(BUFX(idx) & E_IMASK);
                a = (i == 0) ? n - 1 : i - 1;
            }
            tb[k] = T[a];
        }
    }
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int idx = w * 1024 + k * 64 + lane;
        const int c = cls[k];
        u32 i = 0;
        if (c != CLS_NONE) {
            const u64 e_ = BUFX(idx);
            i = (u32)(e_ & E_IMASK);
            // the first sub-group of a split group keeps its rank: no write
            if (MODE == MODE_ROUND0 ? (i >= rlo && i < rhi) : (MODE == MODE_NORMAL && ngrp[k] != (u32)(e_ >> 40))) rank[i] = ngrp[k];
            if (MODE == MODE_FINAL) bwt[spos[k]] = tb[k];
        }
        if (MODE != MODE_FINAL) {
            u64 ms = __ballot(c == CLS_SMALL), ml = __ballot(c == CLS_LKEEP);
            const u64 lt = (1ull << lane) - 1ull;
            // list traffic is streaming: keep it from evicting the rank / BWT lines being scattered into
            if (c == CLS_SMALL) __builtin_nontemporal_store(((u64)ngrp[k] << 40) | i, &sOut[sRun + __popcll(ms & lt)]);
            if (c == CLS_LKEEP) __builtin_nontemporal_store(((u64)ngrp[k] << 40) | i, &lOut[lRun + __popcll(ml & lt)]);
            sRun += __popcll(ms);
            lRun += __popcll(ml);
        }
    }
}

// ------------------------------------------------------------------ run rule in round 0
// The run rule (below, "long repeats at a fixed distance") needs nothing a doubling round produces:
// a group of the round-0 list -- equal first d0 symbols, members in index order -- whose members are
// i0, i0 + p, i0 + 2p, ... inside ONE recorded region of distance p > d0 is ordered by T alone (every
// pair (i, i + p) first differs at the region's end e, T[e - p] against T[e]).  Such a group is
// finished in the re-rank of round 0: its members get their final ranks (head + slot, or the
// reverse) and never enter a list, so the first doubling round -- the largest -- does not gather,
// sort and re-rank them.  Only groups that lie inside one tile are taken (the flags of a group are
// OR-ed at its head slot in LDS); a group that crosses a tile boundary takes the ordinary path, as
// does anything that fails a check.  Phase 1 and phase 3 of the re-rank both apply it, so their
// counts agree.
#define GF_R0_RUN 16u
#define GF_R0_ASC 2u
#define GF_R0_DESC 4u
#define GF_R0_BAD 8u
__device__ __forceinline__ void round0_run_rule(const u64 *buf, int cnt, int (&cls)[16], u32 (&ngrp)[16],
                                                const u32 (&spos)[16], const int (&gsz)[16], u8 *gflag /* [RTILE] */,
                                                const u32 *s_runs, u32 d0) {
    const int w = wave_id(), lane = lane_id();
    const u32 nr = s_runs[0];
    if (nr == 0) return; // (uniform over the workgroup: s_runs is shared)
    for (int q = threadIdx.x; q < RTILE / 4; q += RT_THREADS) ((u32 *)gflag)[q] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int idx = w * 1024 + k * 64 + lane;
        if ((cls[k] == CLS_SMALL || cls[k] == CLS_LKEEP) && gsz[k] > 1) {
            const int slot = (int)(spos[k] - ngrp[k]);
            if (slot >= 1) {
                const u64 e = buf[idx], ep = buf[idx - 1];
                const u32 i1 = (u32)(e & E_IMASK), i0 = (u32)(ep & E_IMASK);
                const u32 d = i1 - i0;
                u32 f = GF_R0_BAD;
                if (i1 > i0 && d > d0) {
                    u32 ok = 0;
                    for (u32 r = 0; r < nr; r++) {
                        const u32 *R = s_runs + 4 + 4 * r;
                        if (R[0] == d && i0 >= R[1] && i1 <= R[2]) ok = R[3];
                    }
                    if (ok) f = GF_R0_RUN | ok;
                }
                if (slot >= 2 && i0 - (u32)(buf[idx - 2] & E_IMASK) != d) f |= GF_R0_BAD;
                const int h = idx - slot;
                atomicOr((u32 *)gflag + (h >> 2), f << ((h & 3) * 8));
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int idx = w * 1024 + k * 64 + lane;
        if ((cls[k] == CLS_SMALL || cls[k] == CLS_LKEEP) && gsz[k] > 1) {
            const int slot = (int)(spos[k] - ngrp[k]);
            const u32 gf = gflag[idx - slot];
            const u32 dir = gf & (GF_R0_ASC | GF_R0_DESC);
            if ((gf & GF_R0_RUN) && !(gf & GF_R0_BAD) && (dir == GF_R0_ASC || dir == GF_R0_DESC)) {
                cls[k] = CLS_SINGLE;
                ngrp[k] += (dir == GF_R0_ASC) ? (u32)slot : (u32)(gsz[k] - 1 - slot);
            }
        }
    }
}

// ------------------------------------------------------------------ L-path re-rank
// phase 1: per tile {last old head + 1, last new head + 1, LKEEP count}
template <int MODE>
__global__ __launch_bounds__(RT_THREADS) void k2_lrerank_p1(const u64 *__restrict__ key,
                                                            const u32 *__restrict__ cntArr,
                                                            const u32 *__restrict__ state, u32 want,
                                                            u32 *__restrict__ tileagg, u32 S, u32 TPB, u32 gtiles, u32 gnb,
                                                            const u32 *__restrict__ huse, const u32 *__restrict__ runs) {
    XCD_MAP(gtiles, gnb)
    const u32 b = bz_b;
    if (state[b] != want) return;
    const u32 m = cntArr[b];
    const u32 base = bz_t * RTILE;
    if (base >= m) return;
    __shared__ u64 buf[RTILE];
    __shared__ int sm[16];
    __shared__ __attribute__((aligned(16))) u8 gflag[MODE == MODE_ROUND0 ? RTILE : 4];
    __shared__ u32 s_runs[MODE == MODE_ROUND0 ? RUNS_STRIDE : 1];
    if (MODE == MODE_ROUND0 && threadIdx.x < RUNS_STRIDE)
        s_runs[threadIdx.x] = runs ? runs[(size_t)b * RUNS_STRIDE + threadIdx.x] : 0u;
    const u64 *kp = key + (size_t)b * S;
    const int cnt = (int)min((u32)RTILE, m - base);
    {
        // the slot is padded to whole tiles: read the tile unconditionally, 16 loads in flight
        u64 tmp[RT_ITEMS];
#pragma unroll
        for (int k = 0; k < RT_ITEMS; k++) tmp[k] = __builtin_nontemporal_load(&kp[base + k * RT_THREADS + threadIdx.x]);
#pragma unroll
        for (int k = 0; k < RT_ITEMS; k++) {
            const int q = k * RT_THREADS + threadIdx.x;
            buf[q] = (q < cnt) ? tmp[k] : ~0ull;
        }
    }
    const bool has_prev = base > 0;
    const u64 prevE = has_prev ? kp[base - 1] : 0;
    bool end_head = true;
    if (base + cnt < m) end_head = (kp[base + cnt] >> KEY_SHIFT) != (kp[base + cnt - 1] >> KEY_SHIFT);
    __syncthreads();
    int cls[16];
    u32 ngrp[16], spos[16];
    int nheads;
    // carries are irrelevant for the aggregates: use NOHEAD so that heads before the tile read < 0
    int gsz[16];
    tile_classify<MODE>(buf, cnt, prevE, has_prev, NOHEAD, NOHEAD, end_head, sm, cls, ngrp, spos, nheads,
                        MODE == MODE_ROUND0 ? gsz : nullptr);
    if (MODE == MODE_ROUND0) round0_run_rule(buf, cnt, cls, ngrp, spos, gsz, gflag, s_runs, huse[b]);
    // aggregates
    const int w = wave_id(), lane = lane_id();
    int lastOld = 0, lastNew = 0;
    u32 keep = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int idx = w * 1024 + k * 64 + lane;
        if (idx < cnt) {
            u64 e = buf[idx];
            u64 ep = (idx == 0) ? prevE : buf[idx - 1];
            bool first = (idx == 0 && !has_prev);
            bool nh = first || (e >> KEY_SHIFT) != (ep >> KEY_SHIFT);
            bool oh = (MODE == MODE_ROUND0) ? first : (first || (e >> 40) != (ep >> 40));
            if (nh) lastNew = (int)base + idx + 1;
            if (oh) lastOld = (int)base + idx + 1;
        }
        keep += (cls[k] == CLS_LKEEP);
    }
    __shared__ u32 red[8];
    u32 tO = block_reduce_max<RT_THREADS>((u32)lastOld, red);
    u32 tN = block_reduce_max<RT_THREADS>((u32)lastNew, red);
    u32 tK = block_reduce_add<RT_THREADS>(keep, red);
    if (threadIdx.x == 0) {
        u32 *o = tileagg + ((size_t)b * TPB + bz_t) * 4;
        o[0] = tO; o[1] = tN; o[2] = tK; o[3] = (u32)nheads;
    }
}

// phase 2: per block exclusive scans over tiles (max, max, add); mLn = total keep
__global__ __launch_bounds__(256) void k2_lrerank_p2(const u32 *__restrict__ cntArr,
                                                     const u32 *__restrict__ state, u32 want,
                                                     u32 *__restrict__ tileagg, u32 *__restrict__ mLn,
                                                     u32 *__restrict__ leftg, int final_mode, u32 TPB) {
    const u32 b = blockIdx.x;
    if (state[b] != want) return;
    const u32 tiles = (cntArr[b] + RTILE - 1) / RTILE;
    __shared__ u32 sm[8];
    u32 cO = 0, cN = 0, cK = 0, cH = 0;
    for (u32 t0 = 0; t0 < tiles; t0 += 256) {
        u32 t = t0 + threadIdx.x;
        u32 *a = tileagg + ((size_t)b * TPB + t) * 4;
        u32 vO = 0, vN = 0, vK = 0, vH = 0;
        if (t < tiles) { vO = a[0]; vN = a[1]; vK = a[2]; vH = a[3]; }
        u32 totO, totN, totK, totH;
        u32 iO = block_incl_scan_max<256>(vO, sm, &totO);
        u32 eO = __shfl_up(iO, 1, 64);
        if (lane_id() == 0) { eO = 0; for (int q = 0; q < wave_id(); q++) eO = max(eO, sm[q]); }
        __syncthreads();
        u32 iN = block_incl_scan_max<256>(vN, sm, &totN);
        u32 eN = __shfl_up(iN, 1, 64);
        if (lane_id() == 0) { eN = 0; for (int q = 0; q < wave_id(); q++) eN = max(eN, sm[q]); }
        __syncthreads();
        u32 eK = block_excl_scan_add<256>(vK, sm, &totK);
        __syncthreads();
        u32 eH = block_excl_scan_add<256>(vH, sm, &totH);
        (void)eH;
        if (t < tiles) { a[0] = max(eO, cO); a[1] = max(eN, cN); a[2] = cK + eK; }
        cO = max(cO, totO); cN = max(cN, totN); cK += totK; cH += totH;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (final_mode) atomicAdd(&leftg[b], cH);
        else mLn[b] = cK;
    }
}

// phase 3
template <int MODE>
__global__ __launch_bounds__(RT_THREADS) void k2_lrerank_p3(
    const u64 *__restrict__ key, u64 *__restrict__ lOutAll, u64 *slA, u64 *slB,
    const u32 *__restrict__ par, const u32 *__restrict__ cntArr,
    const u32 *__restrict__ state, u32 want, const u32 *__restrict__ tileagg,
    u32 *__restrict__ rank, const u8 *__restrict__ T, u8 *__restrict__ bwt,
    const u32 *__restrict__ nArr, u32 *__restrict__ mSn, u32 *__restrict__ gcount, u32 S, u32 TPB,
    u32 gtiles, u32 gnb, u32 nsw, const u32 *__restrict__ huse, const u32 *__restrict__ runs) {
    // round 0 writes the rank of every position, in sorted order: a block's rank array does not
    // stay in its XCD's L2 next to the streaming lists, so the pass is swept nsw times, each
    // sweep writing only the ranks that fall in one 1/nsw window of positions
    XCD_MAP(gtiles * nsw, gnb)
    const u32 b = bz_b;
    if (state[b] != want) return;
    const u32 m = cntArr[b];
    const u32 sweep = bz_t / gtiles;
    bz_t -= sweep * gtiles;
    const u32 base = bz_t * RTILE;
    if (base >= m) return;
    __shared__ u64 buf[RTILE];
    __shared__ int sm[16];
    __shared__ __attribute__((aligned(16))) u8 gflag[MODE == MODE_ROUND0 ? RTILE : 4];
    __shared__ u32 s_runs[MODE == MODE_ROUND0 ? RUNS_STRIDE : 1];
    if (MODE == MODE_ROUND0 && threadIdx.x < RUNS_STRIDE)
        s_runs[threadIdx.x] = runs ? runs[(size_t)b * RUNS_STRIDE + threadIdx.x] : 0u;
    const u64 *kp = key + (size_t)b * S;
    const int cnt = (int)min((u32)RTILE, m - base);
    {
        // the slot is padded to whole tiles: read the tile unconditionally, 16 loads in flight
        u64 tmp[RT_ITEMS];
#pragma unroll
        for (int k = 0; k < RT_ITEMS; k++) tmp[k] = __builtin_nontemporal_load(&kp[base + k * RT_THREADS + threadIdx.x]);
#pragma unroll
        for (int k = 0; k < RT_ITEMS; k++) {
            const int q = k * RT_THREADS + threadIdx.x;
            buf[q] = (q < cnt) ? tmp[k] : ~0ull;
        }
    }
    const bool has_prev = base > 0;
    const u64 prevE = has_prev ? kp[base - 1] : 0;
    bool end_head = true;
    if (base + cnt < m) end_head = (kp[base + cnt] >> KEY_SHIFT) != (kp[base + cnt - 1] >> KEY_SHIFT);
    const u32 *ag = tileagg + ((size_t)b * TPB + bz_t) * 4;
    int carryOld = (MODE == MODE_ROUND0) ? -(int)base : (ag[0] ? (int)ag[0] - 1 - (int)base : NOHEAD);
    int carryNew = ag[1] ? (int)ag[1] - 1 - (int)base : NOHEAD;
    const u32 lBase = ag[2];
    __syncthreads();
    int cls[16];
    u32 ngrp[16], spos[16];
    int nheads;
    int gsz[16];
    tile_classify<MODE>(buf, cnt, prevE, has_prev, carryOld, carryNew, end_head, sm, cls, ngrp, spos, nheads,
                        MODE == MODE_ROUND0 ? gsz : nullptr);
    if (MODE == MODE_ROUND0) round0_run_rule(buf, cnt, cls, ngrp, spos, gsz, gflag, s_runs, huse[b]);
    // S survivors go to the buffer the S-kernel of this round also appends to
    u64 *sOut = (par[b] ? slA : slB) + (size_t)b * S;
    const u32 nblk = nArr[b], win = (nblk + nsw - 1) / nsw;
    tile_emit<MODE>(buf, cnt, cls, ngrp, spos, rank + (size_t)b * S, T + (size_t)b * S,
                    bwt + (size_t)b * S, nblk, sOut, mSn + b, lOutAll + (size_t)b * S, lBase, sm, gcount + b,
                    nsw > 1 ? sweep * win : 0u, nsw > 1 ? (sweep + 1) * win : ~0u, sweep == 0);
}

// ------------------------------------------------------------------ round-0 re-rank, 16 consecutive elements per thread
// The same classification and outputs as k2_lrerank_p1 / p3<MODE_ROUND0>, laid out the other way:
// a thread owns 16 consecutive slots of the tile (read from the swizzled LDS copy) and walks them
// with "slot of my group's head" / "slot of the next head" in registers; what enters a thread is a
// max-scan (last head before) and a suffix min-scan (first head behind) over the threads.  The
// row-per-wave-step layout pays ballots and 64-bit mask arithmetic per row of 64 (75 vector
// instructions per row in tile_classify alone); here the cross-lane work is paid per 1024 slots.
struct R0Cls {
    u64 e[16];
    int hd[16];  // tile-local slot of the group head (negative: the group entered from an earlier tile)
    int ohd[16]; // (doubling rounds only) tile-local slot of the head of the element's OLD group
    int nxt[16]; // tile-local slot of the next head behind the element (cnt, or CROSS)
    u32 heads;   // bit k: element k is a head
    u32 valid;   // bit k: element k exists
    u32 cls;     // 2 bits per element: CLS_*
};

// MODE_ROUND0: one kind of head (the key changes).  MODE_NORMAL (doubling rounds): old heads (grp changes) as
// well, for the rank base of an element: its old group's rank + the distance of its new head from the old one.
template <bool CLASSES, int MODE = MODE_ROUND0>
__device__ __forceinline__ void r0_classify(const u64 *buf, int cnt, u64 prevE, bool has_prev, int carryNew,
                                            bool end_head, int *sm /* 16 ints */, R0Cls &R, u64 &pe,
                                            int &nheads, int &lastHeadTile, int carryOld = NOHEAD,
                                            int *lastOldTile = nullptr) {
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const int q0 = tid * 16;
#pragma unroll
    for (int k = 0; k < 16; k++) R.e[k] = buf[SZ(q0 + k)];
    pe = tid ? buf[SZ(tid ? q0 - 1 : 0)] : prevE; // the element before the thread's first
    u32 heads = 0, valid = 0, oheads = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const bool v = q0 + k < cnt;
        const u64 ep = k ? R.e[k ? k - 1 : 0] : pe;
        const bool first = k == 0 && tid == 0 && !has_prev;
        const bool h = first || (R.e[k] >> KEY_SHIFT) != (ep >> KEY_SHIFT);
        heads |= (h && v) ? 1u << k : 0u;
        valid |= v ? 1u << k : 0u;
        if (MODE == MODE_NORMAL) {
            const bool oh = first || (R.e[k] >> 40) != (ep >> 40);
            oheads |= (oh && v) ? 1u << k : 0u;
        }
    }
    R.heads = heads;
    R.valid = valid;
    u32 oinc = 0;
    if (MODE == MODE_NORMAL) oinc = wave_incl_scan_max(oheads ? (u32)q0 + (31u - (u32)__clz(oheads)) + 1u : 0u);
    const u32 lastp1 = heads ? (u32)q0 + (31u - (u32)__clz(heads)) + 1u : 0u;
    const u32 firstv = heads ? (u32)q0 + (u32)__ffs(heads) - 1u : 0xffffffffu;
    const u32 inc = wave_incl_scan_max(lastp1);
    u32 suf = firstv; // inclusive suffix minimum over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 o = __shfl_down(suf, d, 64);
        if (lane + d < 64) suf = min(suf, o);
    }
    const u32 hc = wave_reduce_add((u32)__popc(heads));
    __syncthreads();
    if (lane == 63) {
        sm[w] = (int)inc;
        if (MODE == MODE_NORMAL) sm[12 + w] = (int)oinc;
    }
    if (lane == 0) {
        sm[4 + w] = (int)suf;
        sm[8 + w] = (int)hc;
    }
    __syncthreads();
    u32 before = 0;
    for (int q = 0; q < w; q++) before = max(before, (u32)sm[q]);
    const u32 up = __shfl_up(inc, 1, 64);
    if (lane > 0) before = max(before, up);
    if (MODE == MODE_NORMAL) {
        u32 obefore = 0;
        for (int q = 0; q < w; q++) obefore = max(obefore, (u32)sm[12 + q]);
        const u32 oup = __shfl_up(oinc, 1, 64);
        if (lane > 0) obefore = max(obefore, oup);
        int ocur = obefore ? (int)obefore - 1 : carryOld;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if ((oheads >> k) & 1u) ocur = q0 + k;
            R.ohd[k] = ocur;
        }
        if (lastOldTile) *lastOldTile = (int)max(max((u32)sm[12], (u32)sm[13]), max((u32)sm[14], (u32)sm[15])) - 1;
    }
    u32 after = 0xffffffffu;
    for (int q = 3; q > w; q--) after = min(after, (u32)sm[4 + q]);
    const u32 dn = __shfl_down(suf, 1, 64);
    if (lane < 63) after = min(after, dn);
    nheads = sm[8] + sm[9] + sm[10] + sm[11];
    lastHeadTile = (int)max(max((u32)sm[0], (u32)sm[1]), max((u32)sm[2], (u32)sm[3])) - 1;
    int cur = before ? (int)before - 1 : carryNew;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if ((heads >> k) & 1u) cur = q0 + k;
        R.hd[k] = cur;
    }
    if (CLASSES) {
        int nx = after != 0xffffffffu ? (int)after : (end_head ? cnt : CROSS);
        u32 cls = 0;
#pragma unroll
        for (int k = 15; k >= 0; k--) {
            R.nxt[k] = nx;
            int c = CLS_NONE;
            if ((valid >> k) & 1u) {
                if (R.hd[k] < 0 || nx == CROSS) c = CLS_LKEEP;
                else {
                    const int size = nx - R.hd[k];
                    c = (size == 1) ? CLS_SINGLE : (size <= GS ? CLS_SMALL : CLS_LKEEP);
                }
            }
            cls |= (u32)c << (2 * k);
            if ((heads >> k) & 1u) nx = q0 + k;
        }
        R.cls = cls;
    }
}

// round0_run_rule on this layout: groups inside the tile that lie in one recorded region at one
// distance are finished here (class SINGLE, rank = head + slot or its reverse)
__device__ __forceinline__ void r0_rule(const u64 *buf, R0Cls &R, u64 pe, u32 (&rk)[16], u8 *gflag /* [RTILE] */,
                                        const u32 *s_runs, u32 d0) {
    const u32 nr = s_runs[0];
    if (nr == 0) return; // (uniform over the workgroup: s_runs is shared)
    const int q0 = threadIdx.x * 16;
    for (int q = threadIdx.x; q < RTILE / 4; q += RT_THREADS) ((u32 *)gflag)[q] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int c = (int)((R.cls >> (2 * k)) & 3u);
        const int gsz = (R.hd[k] >= 0 && R.nxt[k] != CROSS) ? R.nxt[k] - R.hd[k] : 0;
        if ((c == CLS_SMALL || c == CLS_LKEEP) && gsz > 1) {
            const int slot = q0 + k - R.hd[k];
            if (slot >= 1) {
                const u64 e = R.e[k], ep = k ? R.e[k ? k - 1 : 0] : pe;
                const u32 i1 = (u32)(e & E_IMASK), i0 = (u32)(ep & E_IMASK);
                const u32 d = i1 - i0;
                u32 f = GF_R0_BAD;
                if (i1 > i0 && d > d0) {
                    u32 ok = 0;
                    for (u32 r = 0; r < nr; r++) {
                        const u32 *Rg = s_runs + 4 + 4 * r;
                        if (Rg[0] == d && i0 >= Rg[1] && i1 <= Rg[2]) ok = Rg[3];
                    }
                    if (ok) f = GF_R0_RUN | ok;
                }
                if (slot >= 2) {
                    const u64 epp = k >= 2 ? R.e[k >= 2 ? k - 2 : 0] : (k == 1 ? pe : buf[SZ(q0 + k >= 2 ? q0 + k - 2 : 0)]);
                    if (i0 - (u32)(epp & E_IMASK) != d) f |= GF_R0_BAD;
                }
                const int h = R.hd[k];
                atomicOr((u32 *)gflag + (h >> 2), f << ((h & 3) * 8));
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int c = (int)((R.cls >> (2 * k)) & 3u);
        const int gsz = (R.hd[k] >= 0 && R.nxt[k] != CROSS) ? R.nxt[k] - R.hd[k] : 0;
        if ((c == CLS_SMALL || c == CLS_LKEEP) && gsz > 1) {
            const int slot = q0 + k - R.hd[k];
            const u32 gf = gflag[R.hd[k]];
            const u32 dir = gf & (GF_R0_ASC | GF_R0_DESC);
            if ((gf & GF_R0_RUN) && !(gf & GF_R0_BAD) && (dir == GF_R0_ASC || dir == GF_R0_DESC)) {
                R.cls = (R.cls & ~(3u << (2 * k))) | ((u32)CLS_SINGLE << (2 * k));
                rk[k] += (dir == GF_R0_ASC) ? (u32)slot : (u32)(gsz - 1 - slot);
            }
        }
    }
}

// loads the tile into the swizzled LDS copy (16 loads in flight; the slot is padded to whole tiles)
__device__ __forceinline__ void r0_tile_to_lds(const u64 *__restrict__ kp, u32 base, int cnt, u64 *buf) {
    u64 tmp[RT_ITEMS];
#pragma unroll
    for (int k = 0; k < RT_ITEMS; k++) tmp[k] = __builtin_nontemporal_load(&kp[base + k * RT_THREADS + threadIdx.x]);
#pragma unroll
    for (int k = 0; k < RT_ITEMS; k++) {
        const int q = k * RT_THREADS + threadIdx.x;
        buf[SZ(q)] = (q < cnt) ? tmp[k] : ~0ull;
    }
}

// phase 1: per tile {last old head + 1, last new head + 1, LKEEP count, heads}
__global__ __launch_bounds__(RT_THREADS) void k2_r0rank_p1(const u64 *__restrict__ key,
                                                           const u32 *__restrict__ cntArr,
                                                           const u32 *__restrict__ state, u32 want,
                                                           u32 *__restrict__ tileagg, u32 S, u32 TPB, u32 gtiles, u32 gnb,
                                                           const u32 *__restrict__ huse, const u32 *__restrict__ runs) {
    XCD_MAP(gtiles, gnb)
    const u32 b = bz_b;
    if (state[b] != want) return;
    const u32 m = cntArr[b];
    const u32 base = bz_t * RTILE;
    if (base >= m) return;
    __shared__ u64 buf[RTILE];
    __shared__ int sm[16];
    __shared__ __attribute__((aligned(16))) u8 gflag[RTILE];
    __shared__ u32 s_runs[RUNS_STRIDE];
    __shared__ u32 red[8];
    if (threadIdx.x < RUNS_STRIDE) s_runs[threadIdx.x] = runs ? runs[(size_t)b * RUNS_STRIDE + threadIdx.x] : 0u;
    const u64 *kp = key + (size_t)b * S;
    const int cnt = (int)min((u32)RTILE, m - base);
    r0_tile_to_lds(kp, base, cnt, buf);
    const bool has_prev = base > 0;
    const u64 prevE = has_prev ? kp[base - 1] : 0;
    bool end_head = true;
    if (base + cnt < m) end_head = (kp[base + cnt] >> KEY_SHIFT) != (kp[base + cnt - 1] >> KEY_SHIFT);
    __syncthreads();
    R0Cls R;
    u64 pe;
    int nheads, lastHeadTile;
    r0_classify<true>(buf, cnt, prevE, has_prev, NOHEAD, end_head, sm, R, pe, nheads, lastHeadTile);
    u32 rk[16];
#pragma unroll
    for (int k = 0; k < 16; k++) rk[k] = 0;
    r0_rule(buf, R, pe, rk, gflag, s_runs, huse[b]);
    u32 keep = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) keep += ((R.cls >> (2 * k)) & 3u) == (u32)CLS_LKEEP;
    const u32 tK = block_reduce_add<RT_THREADS>(keep, red);
    if (threadIdx.x == 0) {
        u32 *o = tileagg + ((size_t)b * TPB + bz_t) * 4;
        o[0] = base == 0 ? 1u : 0u; // (round 0 has no old groups: only the block's first element counts as an old head)
        o[1] = lastHeadTile >= 0 ? base + (u32)lastHeadTile + 1u : 0u;
        o[2] = tK;
        o[3] = (u32)nheads;
    }
}

// phase 3: ranks (one position window per sweep), survivors to the S- and the L-list in slot order
__global__ __launch_bounds__(RT_THREADS) void k2_r0rank_p3(
    const u64 *__restrict__ key, u64 *__restrict__ lOutAll, u64 *slA, u64 *slB,
    const u32 *__restrict__ par, const u32 *__restrict__ cntArr,
    const u32 *__restrict__ state, u32 want, const u32 *__restrict__ tileagg,
    u32 *__restrict__ rank, const u32 *__restrict__ nArr, u32 *__restrict__ mSn, u32 *__restrict__ gcount,
    u32 S, u32 TPB, u32 gtiles, u32 gnb, u32 nsw, const u32 *__restrict__ huse, const u32 *__restrict__ runs) {
    XCD_MAP(gtiles * nsw, gnb)
    const u32 b = bz_b;
    if (state[b] != want) return;
    const u32 m = cntArr[b];
    const u32 sweep = bz_t / gtiles;
    bz_t -= sweep * gtiles;
    const u32 base = bz_t * RTILE;
    if (base >= m) return;
    __shared__ u64 buf[RTILE];
    __shared__ int sm[16];
    __shared__ __attribute__((aligned(16))) u8 gflag[RTILE];
    __shared__ u32 s_runs[RUNS_STRIDE];
    __shared__ u32 red[8];
    __shared__ u32 s_sbase;
    if (threadIdx.x < RUNS_STRIDE) s_runs[threadIdx.x] = runs ? runs[(size_t)b * RUNS_STRIDE + threadIdx.x] : 0u;
    const u64 *kp = key + (size_t)b * S;
    const int cnt = (int)min((u32)RTILE, m - base);
    r0_tile_to_lds(kp, base, cnt, buf);
    const bool has_prev = base > 0;
    const u64 prevE = has_prev ? kp[base - 1] : 0;
    bool end_head = true;
    if (base + cnt < m) end_head = (kp[base + cnt] >> KEY_SHIFT) != (kp[base + cnt - 1] >> KEY_SHIFT);
    const u32 *ag = tileagg + ((size_t)b * TPB + bz_t) * 4;
    const int carryNew = ag[1] ? (int)ag[1] - 1 - (int)base : NOHEAD;
    const u32 lBase = ag[2];
    __syncthreads();
    const bool main_sweep = sweep == 0;
    const bool rule = s_runs[0] != 0;
    R0Cls R;
    u64 pe;
    int nheads, lastHeadTile;
    // a later sweep needs the ranks only; classes (and the rule, which moves ranks) where regions exist
    if (main_sweep || rule) r0_classify<true>(buf, cnt, prevE, has_prev, carryNew, end_head, sm, R, pe, nheads, lastHeadTile);
    else r0_classify<false>(buf, cnt, prevE, has_prev, carryNew, end_head, sm, R, pe, nheads, lastHeadTile);
    u32 rk[16];
#pragma unroll
    for (int k = 0; k < 16; k++) rk[k] = base + (u32)R.hd[k]; // (a group from an earlier tile: its head's position)
    if (rule) r0_rule(buf, R, pe, rk, gflag, s_runs, huse[b]);
    const u32 nblk = nArr[b], win = (nblk + nsw - 1) / nsw;
    const u32 rlo = nsw > 1 ? sweep * win : 0u, rhi = nsw > 1 ? (sweep + 1) * win : ~0u;
    u32 *rb = rank + (size_t)b * S;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const u32 i = (u32)(R.e[k] & E_IMASK);
        if (((R.valid >> k) & 1u) && i >= rlo && i < rhi) rb[i] = rk[k];
    }
    if (!main_sweep) return;
    // survivors: counts, offsets, then through LDS (the tile copy is no longer needed) in slot order
    const int q0 = threadIdx.x * 16;
    u32 cs = 0, cl = 0, ch = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const u32 c = (R.cls >> (2 * k)) & 3u;
        cs += c == (u32)CLS_SMALL;
        cl += c == (u32)CLS_LKEEP;
        ch += (c == (u32)CLS_SMALL || c == (u32)CLS_LKEEP) && R.hd[k] == q0 + k;
    }
    u32 totS, totL;
    u32 so = block_excl_scan_add<RT_THREADS>(cs, red, &totS);
    __syncthreads();
    u32 lo = block_excl_scan_add<RT_THREADS>(cl, red, &totL);
    __syncthreads();
    const u32 totH = block_reduce_add<RT_THREADS>(ch, red);
    if (threadIdx.x == 0) {
        s_sbase = totS ? atomicAdd(mSn + b, totS) : 0u;
        if (totH) atomicAdd(gcount + b, totH);
    }
    __syncthreads(); // every thread has its elements in registers: buf becomes the staging area
    lo += totS;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const u32 c = (R.cls >> (2 * k)) & 3u;
        const u64 v = ((u64)rk[k] << 40) | (R.e[k] & E_IMASK);
        if (c == (u32)CLS_SMALL) buf[so++] = v;
        if (c == (u32)CLS_LKEEP) buf[lo++] = v;
    }
    __syncthreads();
    u64 *sOut = (par[b] ? slA : slB) + (size_t)b * S + s_sbase;
    u64 *lOut = lOutAll + (size_t)b * S + lBase;
    // list traffic is streaming: keep it from evicting the rank lines being scattered into
    for (u32 q = threadIdx.x; q < totS; q += RT_THREADS) __builtin_nontemporal_store(buf[q], &sOut[q]);
    for (u32 q = threadIdx.x; q < totL; q += RT_THREADS) __builtin_nontemporal_store(buf[totS + q], &lOut[q]);
}

// The re-rank of a doubling round's L-list on the same layout (k2_lrerank_p1 / p3<MODE_NORMAL>)
__global__ __launch_bounds__(RT_THREADS) void k2_lrank16_p1(const u64 *__restrict__ key,
                                                            const u32 *__restrict__ cntArr,
                                                            const u32 *__restrict__ state, u32 want,
                                                            u32 *__restrict__ tileagg, u32 S, u32 TPB, u32 gtiles, u32 gnb) {
    XCD_MAP(gtiles, gnb)
    const u32 b = bz_b;
    if (state[b] != want) return;
    const u32 m = cntArr[b];
    const u32 base = bz_t * RTILE;
    if (base >= m) return;
    __shared__ u64 buf[RTILE];
    __shared__ int sm[16];
    __shared__ u32 red[8];
    const u64 *kp = key + (size_t)b * S;
    const int cnt = (int)min((u32)RTILE, m - base);
    r0_tile_to_lds(kp, base, cnt, buf);
    const bool has_prev = base > 0;
    const u64 prevE = has_prev ? kp[base - 1] : 0;
    bool end_head = true;
    if (base + cnt < m) end_head = (kp[base + cnt] >> KEY_SHIFT) != (kp[base + cnt - 1] >> KEY_SHIFT);
    __syncthreads();
    R0Cls R;
    u64 pe;
    int nheads, lastHeadTile, lastOldTile;
    r0_classify<true, MODE_NORMAL>(buf, cnt, prevE, has_prev, NOHEAD, end_head, sm, R, pe, nheads, lastHeadTile, NOHEAD,
                                   &lastOldTile);
    u32 keep = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) keep += ((R.cls >> (2 * k)) & 3u) == (u32)CLS_LKEEP;
    const u32 tK = block_reduce_add<RT_THREADS>(keep, red);
    if (threadIdx.x == 0) {
        u32 *o = tileagg + ((size_t)b * TPB + bz_t) * 4;
        o[0] = lastOldTile >= 0 ? base + (u32)lastOldTile + 1u : 0u;
        o[1] = lastHeadTile >= 0 ? base + (u32)lastHeadTile + 1u : 0u;
        o[2] = tK;
        o[3] = (u32)nheads;
    }
}

// ranks of the elements whose group changed, survivors to the S- and the L-list in slot order: what
// tile_emit<MODE_NORMAL> does, from a thread's 16 consecutive elements, through LDS
__device__ __forceinline__ void ts_emit_normal(u64 *buf, const R0Cls &R, u32 *__restrict__ rank_b,
                                               u64 *__restrict__ sOut_b, u32 *mSn_b, u64 *__restrict__ lOut_b,
                                               u32 lBase, u32 *gcount_b, u32 *red /* 8 */, u32 *s_sbase) {
    const int q0 = threadIdx.x * 16;
    u32 rk[16];
    u32 cs = 0, cl = 0, ch = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const u32 gb = (u32)(R.e[k] >> 40);
        rk[k] = gb + (u32)(R.hd[k] - R.ohd[k]);
        const u32 c = (R.cls >> (2 * k)) & 3u;
        cs += c == (u32)CLS_SMALL;
        cl += c == (u32)CLS_LKEEP;
        ch += (c == (u32)CLS_SMALL || c == (u32)CLS_LKEEP) && R.hd[k] == q0 + k;
        // the first sub-group of a split group keeps its rank: no write
        if (c != (u32)CLS_NONE && rk[k] != gb) rank_b[(u32)(R.e[k] & E_IMASK)] = rk[k];
    }
    u32 totS, totL;
    u32 so = block_excl_scan_add<RT_THREADS>(cs, red, &totS);
    __syncthreads();
    u32 lo = block_excl_scan_add<RT_THREADS>(cl, red, &totL);
    __syncthreads();
    const u32 totH = block_reduce_add<RT_THREADS>(ch, red);
    if (threadIdx.x == 0) {
        *s_sbase = totS ? atomicAdd(mSn_b, totS) : 0u;
        if (totH) atomicAdd(gcount_b, totH);
    }
    __syncthreads(); // every thread has its elements in registers: buf becomes the staging area
    lo += totS;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const u32 c = (R.cls >> (2 * k)) & 3u;
        const u64 v = ((u64)rk[k] << 40) | (R.e[k] & E_IMASK);
        if (c == (u32)CLS_SMALL) buf[so++] = v;
        if (c == (u32)CLS_LKEEP) buf[lo++] = v;
    }
    __syncthreads();
    u64 *sOut = sOut_b + *s_sbase;
    for (u32 q = threadIdx.x; q < totS; q += RT_THREADS) __builtin_nontemporal_store(buf[q], &sOut[q]);
    if (lOut_b)
        for (u32 q = threadIdx.x; q < totL; q += RT_THREADS) __builtin_nontemporal_store(buf[totS + q], &lOut_b[lBase + q]);
}

__global__ __launch_bounds__(RT_THREADS) void k2_lrank16_p3(
    const u64 *__restrict__ key, u64 *__restrict__ lOutAll, u64 *slA, u64 *slB,
    const u32 *__restrict__ par, const u32 *__restrict__ cntArr,
    const u32 *__restrict__ state, u32 want, const u32 *__restrict__ tileagg,
    u32 *__restrict__ rank, u32 *__restrict__ mSn, u32 *__restrict__ gcount,
    u32 S, u32 TPB, u32 gtiles, u32 gnb) {
    XCD_MAP(gtiles, gnb)
    const u32 b = bz_b;
    if (state[b] != want) return;
    const u32 m = cntArr[b];
    const u32 base = bz_t * RTILE;
    if (base >= m) return;
    __shared__ u64 buf[RTILE];
    __shared__ int sm[16];
    __shared__ u32 red[8];
    __shared__ u32 s_sbase;
    const u64 *kp = key + (size_t)b * S;
    const int cnt = (int)min((u32)RTILE, m - base);
    r0_tile_to_lds(kp, base, cnt, buf);
    const bool has_prev = base > 0;
    const u64 prevE = has_prev ? kp[base - 1] : 0;
    bool end_head = true;
    if (base + cnt < m) end_head = (kp[base + cnt] >> KEY_SHIFT) != (kp[base + cnt - 1] >> KEY_SHIFT);
    const u32 *ag = tileagg + ((size_t)b * TPB + bz_t) * 4;
    const int carryOld = ag[0] ? (int)ag[0] - 1 - (int)base : NOHEAD;
    const int carryNew = ag[1] ? (int)ag[1] - 1 - (int)base : NOHEAD;
    const u32 lBase = ag[2];
    __syncthreads();
    R0Cls R;
    u64 pe;
    int nheads, lastHeadTile;
    r0_classify<true, MODE_NORMAL>(buf, cnt, prevE, has_prev, carryNew, end_head, sm, R, pe, nheads, lastHeadTile, carryOld);
    // S survivors go to the buffer the S-kernel of this round also appends to
    ts_emit_normal(buf, R, rank + (size_t)b * S, (par[b] ? slA : slB) + (size_t)b * S, mSn + b,
                   lOutAll + (size_t)b * S, lBase, gcount + b, red, &s_sbase);
}

// ------------------------------------------------------------------ S-path: sort in LDS
// Bitonic sort of buf[0..N) (swizzled indexing, N a power of two >= 64).  The network's
// sub-stages are grouped by the bit they exchange on: [0,4), [4,8), [8,12).  For each group a
// thread loads the 2^W elements that differ only in those bits, runs all the group's sub-stages
// of the current merge phase in registers, and stores them back: 21 LDS round trips for N = 4096
// instead of 78.
template <int W>
__device__ __forceinline__ void bitonic_pass(u64 *buf, int logN, int lo, int kk) {
    const int t = threadIdx.x;
    if (t < ((1 << logN) >> W)) {
        const int low = t & ((1 << lo) - 1), high = t >> lo;
        const int base = low | (high << (lo + W));
        u64 v[1 << W];
#pragma unroll
        for (int m = 0; m < (1 << W); m++) v[m] = buf[SZ(base | (m << lo))];
#pragma unroll
        for (int s = W - 1; s >= 0; s--) {
            if (lo + s < kk) {
#pragma unroll
                for (int m = 0; m < (1 << W); m++) {
                    if (!(m & (1 << s))) {
                        const int idx = base | (m << lo);
                        const bool asc = ((idx >> kk) & 1) == 0;
                        const u64 a = v[m], c = v[m | (1 << s)];
                        const bool sw = (a > c) == asc;
                        v[m] = sw ? c : a;
                        v[m | (1 << s)] = sw ? a : c;
                    }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < (1 << W); m++) buf[SZ(base | (m << lo))] = v[m];
    }
    __syncthreads();
}

__device__ __forceinline__ void bitonic_sort_lds(u64 *buf, int N) {
    const int logN = 31 - __clz(N);
    {
        // merge phases 1..4 only touch bits 0..3: one round trip
        const int t = threadIdx.x;
        if (t < (N >> 4)) {
            const int base = t << 4;
            u64 v[16];
#pragma unroll
            for (int m = 0; m < 16; m++) v[m] = buf[SZ(base | m)];
#pragma unroll
            for (int kk = 1; kk <= 4; kk++) {
#pragma unroll
                for (int s = kk - 1; s >= 0; s--) {
#pragma unroll
                    for (int m = 0; m < 16; m++) {
                        if (!(m & (1 << s))) {
                            const bool asc = (((base | m) >> kk) & 1) == 0;
                            const u64 a = v[m], c = v[m | (1 << s)];
                            const bool sw = (a > c) == asc;
                            v[m] = sw ? c : a;
                            v[m | (1 << s)] = sw ? a : c;
                        }
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < 16; m++) buf[SZ(base | m)] = v[m];
        }
        __syncthreads();
    }
    for (int kk = 5; kk <= logN; kk++) {
        if (kk > 8) {
            switch (logN - 8) {
            case 1: bitonic_pass<1>(buf, logN, 8, kk); break;
            case 2: bitonic_pass<2>(buf, logN, 8, kk); break;
            case 3: bitonic_pass<3>(buf, logN, 8, kk); break;
            default: bitonic_pass<4>(buf, logN, 8, kk); break;
            }
        }
        switch (logN - 4) {
        case 2: bitonic_pass<2>(buf, logN, 4, kk); break;
        case 3: bitonic_pass<3>(buf, logN, 4, kk); break;
        default: bitonic_pass<4>(buf, logN, 4, kk); break;
        }
        bitonic_pass<4>(buf, logN, 0, kk);
    }
}

#define GF_NONUNI 1u
#define GF_ASC 2u
#define GF_DESC 4u
#define GF_NONAP 8u
#define GF_RUN 16u

// ------------------------------------------------------------------ long repeats at a fixed distance
// Doubling needs log2(L/h) rounds to tell apart suffixes i and i+p that agree for L symbols (a
// region of period p, or a second copy of the same data p bytes later): nothing splits until h
// reaches L.  T itself settles such pairs at once: with V = {y : T[y] != T[y-p]}, a maximal gap
// (v1, v2) of V is a region [a, e) = [v1+1-p, v2) on which T[y] = T[y-p], and for every pair
// (i, i+p) with a <= i, i+p <= e the suffixes first differ at e: suffix i reads T[e-p], suffix
// i+p reads T[e].  The first S-round votes for the distances p seen between group neighbours,
// k2_pick keeps the two most frequent per block, k2_viol/k2_runs list their long regions, and
// k2_ssort then finishes any group whose index-ordered members are all such pairs (run rule).

// Votes of round 0: the sorted round-0 list (key << 20 | i, equal keys in index order) is sampled at
// the first 64 slots of every fourth tile; neighbours with equal keys vote for their distance (one
// atomic per stretch of equal distances).  The regions picked from these votes serve every round.
__global__ __launch_bounds__(64) void k2_vote0(const u64 *__restrict__ sorted, const u32 *__restrict__ nArr,
                                               const u32 *__restrict__ huse, u32 *__restrict__ pvote,
                                               u32 S, u32 gtiles, u32 gnb) {
    XCD_MAP(gtiles, gnb)
    const u32 b = bz_b;
    const u32 n = nArr[b];
    const u32 t0 = bz_t * (4u * RTILE); // every fourth tile: the picks need counts, not a census
    if (t0 + 1 >= n) return;
    const int lane = lane_id();
    const u64 *in = sorted + (size_t)b * S;
    const u32 q = t0 + 1u + (u32)lane;
    u32 vd = 0;
    if (q < n) {
        const u64 e = in[q], ep = in[q - 1];
        if ((e >> KEY_SHIFT) == (ep >> KEY_SHIFT)) {
            const u32 i1 = (u32)(e & E_IMASK), i0 = (u32)(ep & E_IMASK);
            if (i1 > i0 && i1 - i0 > huse[b]) vd = i1 - i0;
        }
    }
    const u32 vprev = __shfl_up(vd, 1, 64);
    const bool vote = vd != 0;
    const bool rh = vote && (lane == 0 || vprev != vd);
    const u64 hm_ = __ballot(rh), vm_ = __ballot(vote);
    if (rh) {
        const u64 above = (lane == 63) ? 0ull : ((hm_ | ~vm_) & (~0ull << (lane + 1)));
        const u32 cnt = (above ? (u32)__ffsll((unsigned long long)above) - 1u : 64u) - (u32)lane;
        u32 *pv = pvote + ((size_t)b * VT + VSLOT(vd)) * 2;
        atomicAdd(pv, cnt);
        atomicMax(pv + 1, (cnt << 20) | vd);
    }
}

__global__ __launch_bounds__(256) void k2_pick(const u32 *__restrict__ pvote,
                                               const u32 *__restrict__ nArr, u32 *__restrict__ cand,
                                               u32 *__restrict__ runs, u32 nb) {
    const u32 b = blockIdx.x;
    __shared__ u32 sm[8];
    const u32 *pv = pvote + (size_t)b * VT * 2;
    const u32 n = nArr[b];
    u32 first = 0;
    for (int pass = 0; pass < NCAND; pass++) {
        u32 best = 0;
        for (u32 sl = threadIdx.x; sl < VT; sl += 256) {
            u32 cnt = pv[2 * sl];
            if (cnt >= 8 && !(pass == 1 && sl == (first & (VT - 1)) && first))
                best = max(best, (min(cnt, 0x7FFFFu) << 13) | sl);
        }
        best = block_reduce_max<256>(best, sm);
        if (pass == 0) first = best;
        if (threadIdx.x == 0) {
            u32 pp = best ? (pv[2 * (best & (VT - 1)) + 1] & 0xFFFFFu) : 0;
            if (pp == 0 || pp >= n) pp = 0;
            cand[b * NCAND + pass] = pp;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) runs[(size_t)b * RUNS_STRIDE] = 0;
}

// per tile of y in [p, n + p): smallest and largest y with T[y mod n] != T[y - p]
__global__ __launch_bounds__(256) void k2_viol(const u8 *__restrict__ T, const u32 *__restrict__ nArr,
                                               const u32 *__restrict__ cand, u32 *__restrict__ vagg,
                                               u32 S, u32 TPB) {
    const u32 b = blockIdx.z, cidx = blockIdx.y, t = blockIdx.x;
    const u32 p = cand[b * NCAND + cidx];
    if (!p) return;
    const u32 n = nArr[b];
    const u32 base = t * RTILE;
    if (base >= n) return;
    __shared__ u32 sm[8];
    const u8 *Tb = T + (size_t)b * S;
    u32 lo = 0xFFFFFFFFu, hi = 0;
#pragma unroll 4
    for (int k = 0; k < RT_ITEMS; k++) {
        u32 off = base + k * 256 + threadIdx.x;
        if (off < n) {
            u32 y = off + p; // y - p = off
            u32 ym = y >= n ? y - n : y;
            if (Tb[ym] != Tb[off]) {
                lo = min(lo, y);
                hi = max(hi, y);
            }
        }
    }
    hi = block_reduce_max<256>(hi, sm);
    lo = ~block_reduce_max<256>(~lo, sm);
    if (threadIdx.x == 0) {
        u32 *o = vagg + (((size_t)b * NCAND + cidx) * TPB + t) * 2;
        o[0] = lo;
        o[1] = hi;
    }
}

__global__ void k2_runs(const u8 *__restrict__ T, const u32 *__restrict__ nArr,
                        const u32 *__restrict__ cand, const u32 *__restrict__ vagg,
                        u32 *__restrict__ runs, u32 S, u32 TPB, u32 nb) {
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const u32 n = nArr[b];
    const u8 *Tb = T + (size_t)b * S;
    u32 *R = runs + (size_t)b * RUNS_STRIDE;
    u32 cnt = 0;
    for (u32 cidx = 0; cidx < NCAND; cidx++) {
        const u32 p = cand[b * NCAND + cidx];
        if (!p) continue;
        const u32 tiles = (n + RTILE - 1) / RTILE;
        u32 lastv = p - 1, kept = 0; // y < p cannot be compared: the block start bounds the first region
        for (u32 t = 0; t < tiles && kept < MAXRUNS; t++) {
            const u32 *a = vagg + (((size_t)b * NCAND + cidx) * TPB + t) * 2;
            const u32 lo = a[0], hi = a[1];
            if (lo == 0xFFFFFFFFu) continue;
            if (lo - lastv - 1 >= MINRUN) {
                const u32 e = lo, em = e >= n ? e - n : e;
                u32 *o = R + 4 + 4 * cnt;
                o[0] = p;
                o[1] = lastv + 1 - p;
                o[2] = e;
                o[3] = (Tb[e - p] < Tb[em]) ? GF_ASC : GF_DESC;
                cnt++;
                kept++;
            }
            lastv = hi;
        }
    }
    R[0] = cnt;
}


// ------------------------------------------------------------------ long-repeat rules on the L-list
// Same rules as in k2_ssort (arithmetic progression with p <= h, run rule) for groups that are too
// large for an S-window.  The L-list (keyA, k2 already gathered) is grouped and in index order
// inside groups; a group may span tiles, so its flags are OR-ed in three steps:
//   lap_p1  per tile: first/last head, flags of the elements before the first head and from the
//           last head on
//   lap_p2  per block: walk the tiles forward and backward -> for every tile the complete flags
//           and head position of the group entering it and of the group still open at its end
//   lap_p3  per tile: groups that qualify get k2 := slot (or its complement); the radix sort and
//           re-rank of this round then finish them as singletons.
__device__ __forceinline__ u32 lap_pair_flags(u64 e, u64 ep, bool has_pp, u64 epp, u32 hcur,
                                              const u32 *s_runs) {
    u32 f = 0;
    const u32 k2 = (u32)(e >> KEY_SHIFT) & 0xFFFFFu, k2p = (u32)(ep >> KEY_SHIFT) & 0xFFFFFu;
    if (k2 != k2p) f |= (k2 > k2p) ? (GF_NONUNI | GF_ASC) : (GF_NONUNI | GF_DESC);
    const u32 i1 = (u32)(e & E_IMASK), i0 = (u32)(ep & E_IMASK);
    const u32 d = i1 - i0;
    if (i1 <= i0) f |= GF_NONAP;
    else if (d > hcur) {
        u32 ok = 0;
        const u32 nr = s_runs[0];
        for (u32 r = 0; r < nr; r++) {
            const u32 *R = s_runs + 4 + 4 * r;
            if (R[0] == d && i0 >= R[1] && i1 <= R[2]) ok = R[3];
        }
        f |= ok ? (GF_RUN | ok) : GF_NONAP;
    }
    if (has_pp && (epp >> 40) == (e >> 40) && i0 - (u32)(epp & E_IMASK) != d) f |= GF_NONAP;
    return f;
}

__device__ __forceinline__ bool lap_qualifies(u32 gf) {
    const u32 dir = gf & (GF_ASC | GF_DESC);
    return (gf & (GF_NONUNI | GF_RUN)) && !(gf & GF_NONAP) && (dir == GF_ASC || dir == GF_DESC);
}

#define LAP_LOAD()                                                                              \
    const u64 *list = keyA + (size_t)b * S;                                                     \
    u64 ev[RT_ITEMS];                                                                           \
    u32 fl[RT_ITEMS];                                                                           \
    u64 hm[RT_ITEMS];                                                                           \
    u64 e1[RT_ITEMS], e2[RT_ITEMS];                                                             \
    _Pragma("unroll") for (int k = 0; k < RT_ITEMS; k++) {                                      \
        const u32 j = base + (u32)(w * 1024 + k * 64 + lane);                                   \
        const u32 jc = j < m ? j : base;                                                        \
        ev[k] = list[jc];                                                                       \
        e1[k] = list[jc >= 1 ? jc - 1 : 0];                                                     \
        e2[k] = list[jc >= 2 ? jc - 2 : 0];                                                     \
    }                                                                                           \
    _Pragma("unroll") for (int k = 0; k < RT_ITEMS; k++) {                                      \
        const u32 j = base + (u32)(w * 1024 + k * 64 + lane);                                   \
        bool head = false;                                                                      \
        u32 f = 0;                                                                              \
        u64 e = 0;                                                                              \
        if (j < m) {                                                                            \
            e = ev[k];                                                                          \
            if (j == 0) head = true;                                                            \
            else {                                                                              \
                const u64 ep = e1[k];                                                           \
                head = (e >> 40) != (ep >> 40);                                                 \
                if (!head) f = lap_pair_flags(e, ep, j >= 2, e2[k], hcur, s_runs);              \
            }                                                                                   \
        }                                                                                       \
        ev[k] = e;                                                                              \
        fl[k] = f;                                                                              \
        hm[k] = __ballot(head);                                                                 \
    }

__global__ __launch_bounds__(RT_THREADS) void k2_lap_p1(const u64 *__restrict__ keyA,
                                                        const u32 *__restrict__ mL,
                                                        const u32 *__restrict__ state,
                                                        const u32 *__restrict__ huse,
                                                        const u32 *__restrict__ runs,
                                                        u32 *__restrict__ pvote,
                                                        u32 *__restrict__ lagg, u32 S, u32 TPB,
                                                        u32 gtiles, u32 gnb) {
    XCD_MAP(gtiles, gnb)
    const u32 b = bz_b;
    if (state[b] != ST_ACTIVE) return;
    const u32 m = mL[b];
    const u32 base = bz_t * RTILE;
    if (base >= m) return;
    __shared__ u32 s_runs[RUNS_STRIDE];
    __shared__ u32 sm[8];
    __shared__ u32 s_or[2];
    if (threadIdx.x < RUNS_STRIDE) s_runs[threadIdx.x] = runs ? runs[(size_t)b * RUNS_STRIDE + threadIdx.x] : 0u;
    if (threadIdx.x < 2) s_or[threadIdx.x] = 0;
    __syncthreads();
    const int w = wave_id(), lane = lane_id();
    const u32 hcur = huse[b];
    LAP_LOAD()
    if (pvote && w == 0) {
        // votes from the first row of the tile, one atomic per stretch of equal distances
        const u32 j = base + (u32)lane;
        u32 vd = 0;
        if (j >= 1 && j < m && !((hm[0] >> lane) & 1ull)) {
            const u64 ep = list[j - 1];
            const u32 i1 = (u32)(ev[0] & E_IMASK), i0 = (u32)(ep & E_IMASK);
            const bool same_k2 = ((ev[0] ^ ep) >> KEY_SHIFT & 0xFFFFFull) == 0;
            if (same_k2 && i1 > i0 && i1 - i0 > hcur) vd = i1 - i0;
        }
        const u32 vprev = __shfl_up(vd, 1, 64);
        const bool vote = vd != 0;
        const bool rh = vote && (lane == 0 || vprev != vd);
        const u64 hm_ = __ballot(rh), vm_ = __ballot(vote);
        if (rh) {
            const u64 above = (lane == 63) ? 0ull : ((hm_ | ~vm_) & (~0ull << (lane + 1)));
            const u32 cnt = (above ? (u32)__ffsll((unsigned long long)above) - 1u : 64u) - (u32)lane;
            u32 *pv = pvote + ((size_t)b * VT + VSLOT(vd)) * 2;
            atomicAdd(pv, cnt);
            atomicMax(pv + 1, (cnt << 20) | vd);
        }
    }
    // first / last head slot of the tile (+1, 0 = none)
    u32 firstH = 0xFFFFFFFFu, lastH = 0;
#pragma unroll
    for (int k = 0; k < RT_ITEMS; k++) {
        if (hm[k]) {
            const u32 rb = (u32)(w * 1024 + k * 64);
            firstH = min(firstH, rb + (u32)__ffsll((unsigned long long)hm[k]) - 1u);
            lastH = max(lastH, rb + 64u - (u32)__clzll((unsigned long long)hm[k]));
        }
    }
    lastH = block_reduce_max<RT_THREADS>(lastH, sm);                // last head slot + 1
    firstH = ~block_reduce_max<RT_THREADS>(~firstH, sm);            // first head slot or ~0
    u32 pre = 0, suf = 0;
#pragma unroll
    for (int k = 0; k < RT_ITEMS; k++) {
        const u32 idx = (u32)(w * 1024 + k * 64 + lane);
        if (idx < firstH) pre |= fl[k];
        if (lastH && idx >= lastH - 1u) suf |= fl[k];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        pre |= __shfl_xor(pre, d, 64);
        suf |= __shfl_xor(suf, d, 64);
    }
    if (lane == 0) {
        if (pre) atomicOr(&s_or[0], pre);
        if (suf) atomicOr(&s_or[1], suf);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 *o = lagg + ((size_t)b * TPB + bz_t) * 4;
        o[0] = (firstH == 0xFFFFFFFFu) ? 0u : firstH + 1u;
        o[1] = lastH;
        o[2] = s_or[0];
        o[3] = s_or[1];
    }
}

__global__ __launch_bounds__(256) void k2_lap_p2(const u32 *__restrict__ mL,
                                                 const u32 *__restrict__ state,
                                                 const u32 *__restrict__ lagg,
                                                 u32 *__restrict__ lres, u32 TPB) {
    const u32 b = blockIdx.x;
    if (state[b] != ST_ACTIVE) return;
    const u32 m = mL[b];
    const u32 tiles = (m + RTILE - 1) / RTILE;
    if (!tiles) return;
    extern __shared__ u32 lds[]; // [tiles][4] in, then [tiles][4] out
    u32 *in = lds, *out = lds + (size_t)TPB * 4;
    for (u32 q = threadIdx.x; q < tiles * 4; q += 256) in[q] = lagg[(size_t)b * TPB * 4 + q];
    __syncthreads();
    if (threadIdx.x == 0) {
        // forward: flags gathered so far by the group that enters tile t, and its head position
        u32 carryF = 0, carryHead = 0;
        for (u32 t = 0; t < tiles; t++) {
            out[t * 4 + 0] = carryHead;
            out[t * 4 + 1] = carryF; // completed below
            if (in[t * 4 + 1]) {
                carryF = in[t * 4 + 3];
                carryHead = t * RTILE + in[t * 4 + 1] - 1u;
            } else {
                carryF |= in[t * 4 + 2];
            }
        }
        // backward: what the entering group still collects from tile t onward
        u32 after = 0; // of tile t + 1
        for (u32 t = tiles; t-- > 0;) {
            const bool hasHead = in[t * 4 + 1] != 0;
            out[t * 4 + 2] = in[t * 4 + 3] | after;                  // group open at the end of t
            const u32 mine = hasHead ? in[t * 4 + 2] : (in[t * 4 + 2] | after);
            out[t * 4 + 1] |= mine;                                  // group entering t
            after = mine;
        }
    }
    __syncthreads();
    for (u32 q = threadIdx.x; q < tiles * 4; q += 256) lres[(size_t)b * TPB * 4 + q] = out[q];
}

__global__ __launch_bounds__(RT_THREADS) void k2_lap_p3(u64 *__restrict__ keyA,
                                                        const u32 *__restrict__ mL,
                                                        const u32 *__restrict__ state,
                                                        const u32 *__restrict__ huse,
                                                        const u32 *__restrict__ runs,
                                                        const u32 *__restrict__ lagg,
                                                        const u32 *__restrict__ lres, u32 S, u32 TPB,
                                                        u32 gtiles, u32 gnb) {
    XCD_MAP(gtiles, gnb)
    const u32 b = bz_b;
    if (state[b] != ST_ACTIVE) return;
    const u32 m = mL[b];
    const u32 base = bz_t * RTILE;
    if (base >= m) return;
    __shared__ u32 s_runs[RUNS_STRIDE];
    __shared__ int sm[8];
    __shared__ u32 gfl[RTILE]; // flags of the groups that start in this tile, at their head slot
    if (threadIdx.x < RUNS_STRIDE) s_runs[threadIdx.x] = runs ? runs[(size_t)b * RUNS_STRIDE + threadIdx.x] : 0u;
    for (int q = threadIdx.x; q < RTILE; q += RT_THREADS) gfl[q] = 0;
    __syncthreads();
    const int w = wave_id(), lane = lane_id();
    const u32 hcur = huse[b];
    LAP_LOAD()
    const u32 *ag = lagg + ((size_t)b * TPB + bz_t) * 4;
    const u32 *rs = lres + ((size_t)b * TPB + bz_t) * 4;
    const u32 lastH = ag[1];                      // last head slot + 1 (0 = none)
    const u32 inHead = rs[0], inF = rs[1], openF = rs[2];
    // head slot of every element (NOHEAD = the group entered from an earlier tile)
    int lastHead = NOHEAD;
#pragma unroll
    for (int k = 0; k < RT_ITEMS; k++)
        if (hm[k]) lastHead = w * 1024 + k * 64 + 63 - __clzll((unsigned long long)hm[k]);
    if (lane == 0) sm[w] = lastHead;
    __syncthreads();
    int cO = NOHEAD;
    for (int q = 0; q < w; q++)
        if (sm[q] != NOHEAD) cO = sm[q];
    int ohi[RT_ITEMS];
    const u64 le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
#pragma unroll
    for (int k = 0; k < RT_ITEMS; k++) {
        const int rb = w * 1024 + k * 64;
        const u64 mo = hm[k] & le;
        ohi[k] = mo ? rb + 63 - __clzll((unsigned long long)mo) : cO;
        if (hm[k]) cO = rb + 63 - __clzll((unsigned long long)hm[k]);
        if (fl[k] && ohi[k] != NOHEAD) atomicOr(&gfl[ohi[k]], fl[k]);
    }
    __syncthreads();
    // (neighbour tiles read only grp and i of this tile's elements, which stay as they are)
    u64 *lwr = keyA + (size_t)b * S;
#pragma unroll
    for (int k = 0; k < RT_ITEMS; k++) {
        const u32 idx = (u32)(w * 1024 + k * 64 + lane);
        const u32 j = base + idx;
        if (j >= m) continue;
        u32 gf, headpos;
        if (ohi[k] == NOHEAD) {
            gf = inF;
            headpos = inHead;
        } else {
            gf = (lastH && (u32)ohi[k] == lastH - 1u) ? openF : gfl[ohi[k]];
            headpos = base + (u32)ohi[k];
        }
        if (lap_qualifies(gf)) {
            const u32 slot = j - headpos;
            const u32 code = (gf & GF_ASC) ? slot : (0xFFFFFu - slot);
            lwr[j] = (ev[k] & ~(0xFFFFFull << KEY_SHIFT)) | ((u64)code << KEY_SHIFT);
        }
    }
}

#ifdef EXP_TIMING
__device__ unsigned long long g_phase[8];
__device__ unsigned long long g_hist[4];
extern "C" void bz2hip_debug_phases(unsigned long long *out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 8);
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z));
}
extern "C" void bz2hip_debug_hist(unsigned long long *out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hist), sizeof(unsigned long long) * 4);
    unsigned long long z[4] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_hist), z, sizeof(z));
}
#define PH_MARK(k) do { __syncthreads(); if (threadIdx.x == 0) { unsigned long long t_ = wall_clock64(); atomicAdd(&g_phase[k], t_ - ph_t); ph_t = t_; } } while (0)
#define PH_INIT unsigned long long ph_t = wall_clock64();
#else
#define PH_MARK(k)
#define PH_INIT
#endif
template <int MODE>
__global__ __launch_bounds__(RT_THREADS, 3) void k2_ssort(const u64 *slA, const u64 *slB,
                                                       u64 *slAo, u64 *slBo,
                                                       const u32 *__restrict__ par,
                                                       const u32 *__restrict__ mS,
                                                       const u32 *__restrict__ state, u32 want,
                                                       u32 *__restrict__ rank,
                                                       const u8 *__restrict__ T, u8 *__restrict__ bwt,
                                                       const u32 *__restrict__ nArr,
                                                       u32 *__restrict__ mSn, u32 *__restrict__ leftg,
                                                       const u32 *__restrict__ huse,
                                                       u32 *__restrict__ pvote,
                                                       const u32 *__restrict__ runs,
                                                       u32 *__restrict__ gcount,
                                                       u32 S, u32 gtiles, u32 gnb) {
    XCD_MAP(gtiles, gnb)
    const u32 b = bz_b;
    if (state[b] != want) return;
    const u32 m = mS[b];
    const u32 t0 = bz_t * SC;
    if (t0 >= m) return;
    PH_INIT
    __shared__ u64 buf[RTILE];
    __shared__ int sm[16];
    __shared__ int s_start, s_end;
    __shared__ __attribute__((aligned(16))) u8 gflag[RTILE];
    __shared__ u32 s_runs[RUNS_STRIDE];
    if (MODE == MODE_NORMAL) {
        if (threadIdx.x < RUNS_STRIDE)
            s_runs[threadIdx.x] = runs ? runs[(size_t)b * RUNS_STRIDE + threadIdx.x] : 0u;
    }
    const u64 *in = (par[b] ? slB : slA) + (size_t)b * S;
    u64 *out = (par[b] ? slAo : slBo) + (size_t)b * S;
    const int loaded = (int)min((u32)RTILE, m - t0);
    if (threadIdx.x == 0) {
        s_start = CROSS;
        s_end = loaded;
    }
    __syncthreads();
    // load window, find owned span: groups whose head slot is in [t0, t0 + SC)
    int myStart = CROSS, myEnd = CROSS;
    {
        // 16 window loads in flight (clamped: the window may reach past the slot), then the
        // head test against the left neighbour through LDS
        u64 tmp[RT_ITEMS];
#pragma unroll
        for (int k = 0; k < RT_ITEMS; k++) {
            const int q = k * RT_THREADS + threadIdx.x;
            tmp[k] = __builtin_nontemporal_load(&in[t0 + (q < loaded ? q : 0)]);
        }
        const u64 left = t0 ? in[t0 - 1] : 0ull;
#pragma unroll
        for (int k = 0; k < RT_ITEMS; k++) {
            const int q = k * RT_THREADS + threadIdx.x;
            buf[SZ(q)] = (q < loaded) ? tmp[k] : ~0ull;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < RT_ITEMS; k++) {
            const int q = k * RT_THREADS + threadIdx.x;
            if (q < loaded) {
                bool head;
                if (q == 0) head = (t0 == 0) || ((tmp[k] >> 40) != (left >> 40));
                else head = (tmp[k] >> 40) != (buf[SZ(q - 1)] >> 40);
                if (head) {
                    if (q < SC) myStart = min(myStart, q);
                    else myEnd = min(myEnd, q);
                }
            }
        }
    }
    if (myStart != CROSS) atomicMin(&s_start, myStart);
    if (myEnd != CROSS) atomicMin(&s_end, myEnd);
    __syncthreads();
    const int start = s_start, end = s_end;
    if (start == CROSS) return; // owns nothing
    int cnt = end - start;
    PH_MARK(0);
    const int w = wave_id(), lane = lane_id();
    if (MODE == MODE_NORMAL) {
        // Groups whose k2 is the same for every member cannot split this round: they pass
        // through untouched (rank and grp unchanged).  Only the others are compacted and sorted.
        //
        // Arithmetic-progression rule.  Inside a group the list is in index order.  If the indices
        // are i0, i0+p, i0+2p, ... with p <= h (h = depth the ranks stand for), the h-symbol
        // prefixes they share overlap, so T has period p on [i0, ic + h) and every consecutive pair
        // (i, i+p) first differs at the same position e = first y >= ic + h with T[y] != T[y-p]:
        // suffix i reads T[e-p] there, suffix i+p reads T[e].  The group's order is therefore
        // monotone in the index, and one pair with different k2 gives the direction.  Such a group is
        // finished in this round whatever its length (k2 := slot, or its complement).
        u64 ev[16];
        u64 hm[16];
        u32 fl[16]; // per element: GF_* bits it contributes to its group
        int lastHead = NOHEAD;
        const u32 hcur = huse[b];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int rb = w * 1024 + k * 64;
            const int q = rb + lane;
            bool head = false, diff = false;
            u32 f = 0, vd = 0;
            u64 e = ~0ull;
            if (q >= start && q < end) {
                e = buf[SZ(q)];
                if (q == start) head = true;
                else {
                    u64 ep = buf[SZ(q - 1)];
                    head = (e >> 40) != (ep >> 40);
                    if (!head) {
                        const u32 k2 = (u32)(e >> KEY_SHIFT) & 0xFFFFFu, k2p = (u32)(ep >> KEY_SHIFT) & 0xFFFFFu;
                        diff = k2 != k2p;
                        if (diff) f |= (k2 > k2p) ? (GF_NONUNI | GF_ASC) : (GF_NONUNI | GF_DESC);
                        const u32 i1 = (u32)(e & E_IMASK), i0 = (u32)(ep & E_IMASK);
                        const u32 d = i1 - i0;
                        if (i1 <= i0) f |= GF_NONAP;
                        else if (d > hcur) {
                            // run rule: the pair must lie in a recorded region of distance d
                            u32 ok = 0;
                            const u32 nr = s_runs[0];
                            for (u32 r = 0; r < nr; r++) {
                                const u32 *R = s_runs + 4 + 4 * r;
                                if (R[0] == d && i0 >= R[1] && i1 <= R[2]) ok = R[3];
                            }
                            f |= ok ? (GF_RUN | ok) : GF_NONAP;
                            if (!diff) vd = d; // only pairs that do not split this round vote
                        }
                        if (q - 2 >= start) {
                            u64 epp = buf[SZ(q - 2)];
                            if ((epp >> 40) == (e >> 40) && i0 - (u32)(epp & E_IMASK) != d) f |= GF_NONAP;
                        }
                    }
                }
            }
            if (pvote && k == 0 && w == 0) {
                // votes (first 64 slots of the window only: global atomics are dear): neighbouring
                // lanes of one group carry the same distance; one atomic per such stretch
                const u32 vprev = __shfl_up(vd, 1, 64);
                const bool vote = vd != 0;
                const bool rh = vote && (lane == 0 || vprev != vd);
                const u64 hm_ = __ballot(rh), vm_ = __ballot(vote);
                if (rh) {
                    const u64 above = (lane == 63) ? 0ull : ((hm_ | ~vm_) & (~0ull << (lane + 1)));
                    const u32 cnt = (above ? (u32)__ffsll((unsigned long long)above) - 1u : 64u) - (u32)lane;
                    // count: sum; distance: the one with the longest stretch, then the largest
                    // (both independent of the order the votes arrive in)
                    u32 *pv = pvote + ((size_t)b * VT + VSLOT(vd)) * 2;
                    atomicAdd(pv, cnt);
                    atomicMax(pv + 1, (cnt << 20) | vd);
                }
            }
            ev[k] = e;
            fl[k] = f;
            u64 a = __ballot(head);
            hm[k] = a;
            if (a) lastHead = rb + 63 - __clzll((unsigned long long)a);
        }
        for (int q = threadIdx.x; q < RTILE / 4; q += RT_THREADS) ((u32 *)gflag)[q] = 0;
        if (lane == 0) sm[w] = lastHead;
        PH_MARK(5);
        __syncthreads();
        int cO = NOHEAD;
        for (int q = 0; q < w; q++)
            if (sm[q] != NOHEAD) cO = sm[q];
        int ohi[16];
        const u64 le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int rb = w * 1024 + k * 64;
            u64 mo = hm[k] & le;
            ohi[k] = mo ? rb + 63 - __clzll((unsigned long long)mo) : cO;
            if (hm[k]) cO = rb + 63 - __clzll((unsigned long long)hm[k]);
            if (fl[k])
                atomicOr((u32 *)gflag + (ohi[k] >> 2), fl[k] << ((ohi[k] & 3) * 8));
        }
        PH_MARK(6);
        __syncthreads();
        // counts
        u32 cN = 0, cU = 0, cH = 0; // cH: groups that pass through untouched (they stay listed)
        u32 st2 = 0; // per lane, 2 bits per row: bit 0 = goes to the sort, bit 1 = passes through
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int q = w * 1024 + k * 64 + lane;
            bool in_span = (q >= start && q < end);
            const u32 gf = in_span ? gflag[ohi[k]] : 0u;
            bool nonuni = (gf & GF_NONUNI) != 0;
            const bool onedir = (gf & (GF_ASC | GF_DESC)) == GF_ASC || (gf & (GF_ASC | GF_DESC)) == GF_DESC;
            if ((gf & (GF_NONUNI | GF_RUN)) && !(gf & GF_NONAP) && onedir) {
                nonuni = true; // finished this round even if its k2 is still uniform
                // arithmetic progression: final order = index order (ASC) or its reverse (DESC)
                const u32 slot = (u32)(q - ohi[k]);
                const u32 code = (gf & GF_ASC) ? slot : (0xFFFFFu - slot);
                ev[k] = (ev[k] & ~(0xFFFFFull << KEY_SHIFT)) | ((u64)code << KEY_SHIFT);
            }
            const bool uni = in_span && !nonuni;
            st2 |= ((nonuni ? 1u : 0u) | (uni ? 2u : 0u)) << (2 * k);
            cN += __popcll(__ballot(nonuni));
            const u64 um = __ballot(uni);
            cU += __popcll(um);
            cH += __popcll(um & hm[k]);
        }
        if (lane == 0) {
            sm[4 + w] = (int)cN;
            sm[8 + w] = (int)cU;
            sm[12 + w] = (int)cH;
        }
        __syncthreads();
        PH_MARK(7);
        const int totN = sm[4] + sm[5] + sm[6] + sm[7];
        const int totU = sm[8] + sm[9] + sm[10] + sm[11];
        if (threadIdx.x == 0) {
            const u32 totH = (u32)(sm[12] + sm[13] + sm[14] + sm[15]);
            if (totH) atomicAdd(gcount + b, totH);
        }
        __syncthreads();
        if (threadIdx.x == 0) sm[12] = totU ? (int)atomicAdd(mSn + b, (u32)totU) : 0;
        __syncthreads();
        u32 nRun = 0, uRun = (u32)sm[12];
        for (int q = 0; q < w; q++) {
            nRun += (u32)sm[4 + q];
            uRun += (u32)sm[8 + q];
        }
        const u64 lt = (1ull << lane) - 1ull;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const bool isn = (st2 >> (2 * k)) & 1u, isu = (st2 >> (2 * k + 1)) & 1u;
            const u64 nmask = __ballot(isn), umask = __ballot(isu);
            if (isn) buf[SZ((int)(nRun + __popcll(nmask & lt)))] = ev[k];
            if (isu)
                __builtin_nontemporal_store(ev[k] & ~(0xFFFFFull << KEY_SHIFT), &out[uRun + __popcll(umask & lt)]);
            nRun += __popcll(nmask);
            uRun += __popcll(umask);
        }
        cnt = totN;
        if (cnt == 0) return;
#ifdef EXP_TIMING
        {
            // experiment: largest sorted group of the window, weighted by the sorted count
            int mx = 0;
#pragma unroll
            for (int k = 0; k < 16; k++)
                if ((st2 >> (2 * k)) & 1u) mx = max(mx, w * 1024 + k * 64 + lane - ohi[k] + 1);
            __shared__ int s_mx;
            if (threadIdx.x == 0) s_mx = 0;
            __syncthreads();
            atomicMax(&s_mx, mx);
            __syncthreads();
            if (threadIdx.x == 0) {
                const int m_ = s_mx;
                const int slot = m_ <= 8 ? 0 : m_ <= 32 ? 1 : m_ <= 128 ? 2 : 3;
                atomicAdd(&g_hist[slot], (unsigned long long)cnt);
            }
        }
#endif
        int N = 64;
        while (N < cnt) N <<= 1;
        __syncthreads();
        for (int q = cnt + threadIdx.x; q < N; q += RT_THREADS) buf[SZ(q)] = ~0ull;
        __syncthreads();
        PH_MARK(1);
        bitonic_sort_lds(buf, N);
        PH_MARK(2);
    } else {
        // FINAL: order inside groups is irrelevant; keep the span in place at buf[0..cnt)
        int N = 256;
        while (N < end) N <<= 1;
        for (int q = threadIdx.x; q < N; q += RT_THREADS)
            if (q < start || q >= end) buf[SZ(q)] = ~0ull;
        __syncthreads();
        bitonic_sort_lds(buf, N);
    }
    int cls[16];
    u32 ngrp[16], spos[16];
    int nheads;
    tile_classify<MODE, true>(buf, cnt, 0, false, NOHEAD, NOHEAD, true, sm, cls, ngrp, spos, nheads);
    if (MODE == MODE_FINAL && threadIdx.x == 0) atomicAdd(&leftg[b], (u32)nheads);
    PH_MARK(3);
    tile_emit<MODE, true>(buf, cnt, cls, ngrp, spos, rank + (size_t)b * S, T + (size_t)b * S,
                    bwt + (size_t)b * S, nArr[b], out, mSn + b, (u64 *)nullptr, 0, sm, gcount + b);
    PH_MARK(4);
}

// ------------------------------------------------------------------ small L-lists: the whole L-path in one kernel
// Late rounds leave a few groups of more than GS elements: L-lists of some thousand elements that
// still took 21 launches a round (rules 3, five radix passes of 3, re-rank 3), each a few
// microseconds of work behind a launch that costs more.  When every L-list of the batch fits one
// tile, one workgroup per block does the same work on its list in LDS: long-repeat rules (the tile
// holds every group whole, so no flags cross tiles), a bitonic sort of the element words (grp, k2, i
// = the order the five LSD passes produce), head flags, ranks, survivors to the S- / L-lists.
__global__ __launch_bounds__(RT_THREADS) void k2_lsmall(u64 *__restrict__ keyA, u64 *slA, u64 *slB,
                                                        const u32 *__restrict__ par,
                                                        const u32 *__restrict__ mL,
                                                        const u32 *__restrict__ state,
                                                        u32 *__restrict__ rank, const u8 *__restrict__ T,
                                                        u8 *__restrict__ bwt, const u32 *__restrict__ nArr,
                                                        u32 *__restrict__ mSn, u32 *__restrict__ mLn,
                                                        const u32 *__restrict__ huse,
                                                        const u32 *__restrict__ runs,
                                                        u32 *__restrict__ gcount, u32 S) {
    const u32 b = blockIdx.x;
    if (state[b] != ST_ACTIVE) return;
    const u32 m = mL[b];
    if (m == 0 || m > RTILE) return; // (the host takes this path only when every list fits)
    __shared__ u64 buf[RTILE];
    __shared__ int sm[16];
    __shared__ u32 gfl[RTILE]; // flags of every group, at its head slot
    __shared__ u32 s_runs[RUNS_STRIDE];
    if (threadIdx.x < RUNS_STRIDE) s_runs[threadIdx.x] = runs ? runs[(size_t)b * RUNS_STRIDE + threadIdx.x] : 0u;
    for (int q = threadIdx.x; q < RTILE; q += RT_THREADS) gfl[q] = 0;
    __syncthreads();
    const int w = wave_id(), lane = lane_id();
    const u32 hcur = huse[b];
    const u32 base = 0;
    LAP_LOAD()
    int lastHead = NOHEAD;
#pragma unroll
    for (int k = 0; k < RT_ITEMS; k++)
        if (hm[k]) lastHead = w * 1024 + k * 64 + 63 - __clzll((unsigned long long)hm[k]);
    if (lane == 0) sm[w] = lastHead;
    __syncthreads();
    int cO = NOHEAD;
    for (int q = 0; q < w; q++)
        if (sm[q] != NOHEAD) cO = sm[q];
    int ohi[RT_ITEMS];
    const u64 le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
#pragma unroll
    for (int k = 0; k < RT_ITEMS; k++) {
        const int rb = w * 1024 + k * 64;
        const u64 mo = hm[k] & le;
        ohi[k] = mo ? rb + 63 - __clzll((unsigned long long)mo) : cO;
        if (hm[k]) cO = rb + 63 - __clzll((unsigned long long)hm[k]);
        if (fl[k] && ohi[k] != NOHEAD) atomicOr(&gfl[ohi[k]], fl[k]); // (slot 0 is a head: ohi is never NOHEAD for j < m)
    }
    __syncthreads();
    int N = 64;
    while (N < (int)m) N <<= 1;
#pragma unroll
    for (int k = 0; k < RT_ITEMS; k++) {
        const int idx = w * 1024 + k * 64 + lane;
        if (idx < N) {
            u64 e = ~0ull;
            if ((u32)idx < m) {
                e = ev[k];
                const u32 gf = gfl[ohi[k]];
                if (lap_qualifies(gf)) {
                    const u32 slot = (u32)(idx - ohi[k]);
                    const u32 code = (gf & GF_ASC) ? slot : (0xFFFFFu - slot);
                    e = (e & ~(0xFFFFFull << KEY_SHIFT)) | ((u64)code << KEY_SHIFT);
                }
            }
            buf[SZ(idx)] = e;
        }
    }
    __syncthreads();
    bitonic_sort_lds(buf, N);
    int cls[16];
    u32 ngrp[16], spos[16];
    int nheads;
    tile_classify<MODE_NORMAL, true>(buf, (int)m, 0, false, NOHEAD, NOHEAD, true, sm, cls, ngrp, spos, nheads);
    u64 *sOut = (par[b] ? slA : slB) + (size_t)b * S; // the buffer the S-kernel of this round appends to
    // the list was read whole before anything is written: its survivors go back to the front of it
    tile_emit<MODE_NORMAL, true>(buf, (int)m, cls, ngrp, spos, rank + (size_t)b * S, T + (size_t)b * S,
                                 bwt + (size_t)b * S, nArr[b], sOut, mSn + b, keyA + (size_t)b * S, 0, sm, gcount + b);
    __syncthreads();
    if (threadIdx.x == 0) mLn[b] = (u32)(sm[4] + sm[5] + sm[6] + sm[7]); // LKEEP counts per wave (tile_emit)
}

// ------------------------------------------------------------------ BWT bytes
// bwt[rank[i]] = T[(i - 1) mod n] for every rotation: rank and T are streamed, the 1-byte stores
// land in the block's n-byte BWT array, which stays in the XCD's L2 (XCD-aware mapping) until its
// lines are complete.  Rotations still tied at the end (periodic block) all carry their group's
// rank and the same preceding byte; MODE_FINAL then places them one per slot.
__global__ __launch_bounds__(RT_THREADS) void k2_emit_bwt(const u32 *__restrict__ nArr,
                                                          const u32 *__restrict__ rank,
                                                          const u8 *__restrict__ T, u8 *__restrict__ bwt,
                                                          u32 S, u32 gtiles, u32 gnb) {
    XCD_MAP(gtiles, gnb)
    const u32 b = bz_b;
    const u32 n = nArr[b];
    const u32 base = bz_t * RTILE;
    if (base >= n) return;
    const u32 *rk = rank + (size_t)b * S;
    const u8 *t = T + (size_t)b * S;
    u8 *o = bwt + (size_t)b * S;
    u32 rv[RT_ITEMS];
    u8 tv[RT_ITEMS];
#pragma unroll
    for (int k = 0; k < RT_ITEMS; k++) {
        const u32 i = base + k * RT_THREADS + threadIdx.x;
        const u32 ic = i < n ? i : base;
        rv[k] = __builtin_nontemporal_load(&rk[ic]);
        tv[k] = t[ic == 0 ? n - 1 : ic - 1];
    }
#pragma unroll
    for (int k = 0; k < RT_ITEMS; k++) {
        const u32 i = base + k * RT_THREADS + threadIdx.x;
        if (i < n) o[rv[k]] = tv[k];
    }
}

// ------------------------------------------------------------------ finish
__global__ void k2_finish(const u32 *__restrict__ nArr, const u32 *__restrict__ rank,
                          const u32 *__restrict__ mS, const u32 *__restrict__ mL,
                          const u32 *__restrict__ state, const u32 *__restrict__ leftg,
                          const u8 *__restrict__ T, u8 *__restrict__ bwt,
                          u32 *__restrict__ periodic, u32 *__restrict__ orig, u32 S, u32 nb,
                          int tie_rule) {
    u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    u32 n = nArr[b];
    if (n == 0) { orig[b] = 0; periodic[b] = 1; return; }
    u32 k = 1;
    if (state[b] == ST_FIN && leftg[b] > 0) k = (mS[b] + mL[b]) / leftg[b];
    periodic[b] = k;
    u32 og = rank[(size_t)b * S];
    if (k > 1 && k < n && tie_rule == BZ2HIP_TIE_LAST) og += k - 1;
    if (k == n) og = 0; // all-equal block (S/BZip2DivSufSort.swift:3168-3174,3505)
    orig[b] = og;
}

// symbol-present bitmap from the block bytes (stage entry bz2hip_bwt has no RLE1 in front)
__global__ __launch_bounds__(256) void k2_present(const u8 *__restrict__ T,
                                                  const u32 *__restrict__ nArr,
                                                  u32 *__restrict__ present, u32 S, u32 gtiles, u32 gnb) {
    XCD_MAP(gtiles, gnb)
    const u32 b = bz_b;
    const u32 n = nArr[b];
    const u32 base = bz_t * RTILE;
    if (base >= n) return;
    __shared__ u32 pres[8];
    if (threadIdx.x < 8) pres[threadIdx.x] = 0;
    __syncthreads();
    for (int k = 0; k < RT_ITEMS; k++) {
        u32 i = base + k * 256 + threadIdx.x;
        if (i < n) {
            u32 v = T[(size_t)b * S + i];
            atomicOr(&pres[v >> 5], 1u << (v & 31));
        }
    }
    __syncthreads();
    if (threadIdx.x < 8 && pres[threadIdx.x]) atomicOr(&present[b * 8 + threadIdx.x], pres[threadIdx.x]);
}

void run_present(bz2hip_ctx *c, int nb, u32 maxn) {
    Batch &B = c->batch;
    if (nb == 0 || maxn == 0) return;
    HIPCHK(hipMemsetAsync(B.present, 0, (size_t)nb * 8 * sizeof(u32), c->stream));
    k2_present<<<xcd_grid((maxn + RTILE - 1) / RTILE, nb), 256, 0, c->stream>>>(B.T, B.n, B.present, B.S,
                                                                                   (maxn + RTILE - 1) / RTILE, nb);
    KCHECK();
}

// ------------------------------------------------------------------ driver
// tuning lives in the context (read from the environment once, in bz2hip_create):
// plain_mask: bit set = plain (x-fastest) mapping for that kernel family (default: S-sort only);
// r0bits: key bits of the first sort (multiple of 8)
#define nbarg(nb, bit) ((c->plain_mask & (bit)) ? ((u32)(nb) | 0x80000000u) : (u32)(nb))

// `from_T`: the first pass builds the round-0 keys from the block bytes instead of reading `a`
// passes [first_pass, passes) of the sort; `a` is the input of pass `first_pass`, `bbuf` its output
template <int DB, bool STG>
static void radix_sort_db(bz2hip_ctx *c, int nb, u32 maxcnt, const u32 *cntArr, u64 *a, u64 *bbuf,
                          int passes, bool from_T, int first_pass, bool more_follow, bool dig_ready) {
    Batch &B = c->batch;
    const u32 tiles = (maxcnt + RTILE - 1) / RTILE;
    u64 *kin = a, *kout = bbuf;
    // 8-bit passes hand the next pass its digits as a byte array (B.mtfpos is idle during the sort)
    // (more_follow: a later call continues the sort, so the last pass here leaves digits too;
    // dig_ready: the call before this one did)
    const u8 *digIn = (DB == 8 && c->radix_dig && dig_ready) ? B.mtfpos : nullptr;
    for (int p = first_pass; p < passes; p++) {
        int shift = KEY_SHIFT + DB * p;
        const bool ft = from_T && p == 0;
        u8 *digOut = (DB == 8 && c->radix_dig && (p + 1 < passes || more_follow)) ? B.mtfpos : nullptr;
        {
            ProfScope ps(c, P_RADIX_HIST);
            if (ft)
                k2_radix_hist<true, DB><<<xcd_grid(tiles, nb), RT_THREADS, 0, c->stream>>>(
                    kin, cntArr, B.state, B.tilehist, B.S, B.TPB, shift, tiles, nbarg(nb, 8), B.T, B.present, c->r0bits, nullptr);
            else
                k2_radix_hist<false, DB><<<xcd_grid(tiles, nb), RT_THREADS, 0, c->stream>>>(
                    kin, cntArr, B.state, B.tilehist, B.S, B.TPB, shift, tiles, nbarg(nb, 8), B.T, B.present, c->r0bits, digIn);
            KCHECK();
        }
        {
            ProfScope ps(c, P_RADIX_SCAN);
            k2_radix_scan<DB><<<nb, 256, 0, c->stream>>>(cntArr, B.state, B.tilehist, B.digitbase, B.TPB);
            KCHECK();
        }
        {
            ProfScope ps(c, P_RADIX_SCATTER);
            if (ft)
                k2_radix_scatter<true, DB, STG><<<xcd_grid(tiles, nb), RT_THREADS, 0, c->stream>>>(
                    kin, kout, cntArr, B.state, B.tilehist, B.digitbase, B.S, B.TPB, shift, tiles, nbarg(nb, 8),
                    B.T, B.present, c->r0bits, digOut);
            else
                k2_radix_scatter<false, DB, STG><<<xcd_grid(tiles, nb), RT_THREADS, 0, c->stream>>>(
                    kin, kout, cntArr, B.state, B.tilehist, B.digitbase, B.S, B.TPB, shift, tiles, nbarg(nb, 8),
                    B.T, B.present, c->r0bits, digOut);
            KCHECK();
        }
        digIn = digOut;
        std::swap(kin, kout);
    }
}

// sorts the 40 key bits [59:20]: 5 passes of 8 bits or 4 of 10 (c->radix_bits); the sorted words end
// in `bbuf` when the pass count is odd, in `a` when it is even
static int radix_passes(const bz2hip_ctx *c) { return 40 / (int)c->radix_bits; }
// round 0 sorts r0bits key bits only (BZ2HIP_R0BITS; the default 40 = the whole field)
static int radix_passes0(const bz2hip_ctx *c) { return ((int)c->r0bits + (int)c->radix_bits - 1) / (int)c->radix_bits; }
static void radix_sort5(bz2hip_ctx *c, int nb, u32 maxcnt, const u32 *cntArr, u64 *a, u64 *bbuf,
                        bool from_T = false, int passes = 0, int first_pass = 0, bool more_follow = false,
                        bool dig_ready = false) {
    if (!passes) passes = radix_passes(c);
    if (c->radix_bits == 10) {
        if (c->radix_stage) radix_sort_db<10, true>(c, nb, maxcnt, cntArr, a, bbuf, passes, from_T, first_pass, more_follow, dig_ready);
        else radix_sort_db<10, false>(c, nb, maxcnt, cntArr, a, bbuf, passes, from_T, first_pass, more_follow, dig_ready);
    } else {
        if (c->radix_stage) radix_sort_db<8, true>(c, nb, maxcnt, cntArr, a, bbuf, passes, from_T, first_pass, more_follow, dig_ready);
        else radix_sort_db<8, false>(c, nb, maxcnt, cntArr, a, bbuf, passes, from_T, first_pass, more_follow, dig_ready);
    }
}

template <int MODE>
static void lrerank(bz2hip_ctx *c, int nb, u32 maxcnt, const u32 *cntArr, u32 want, const u64 *sorted,
                    u64 *lout) {
    Batch &B = c->batch;
    const u32 tiles = (maxcnt + RTILE - 1) / RTILE;
    dim3 grid(tiles, nb);
    ProfScope ps(c, P_BWT_RERANK);
    const u32 *r0runs = (MODE == MODE_ROUND0 && c->r0_rule) ? B.runs : nullptr;
    const bool serial16 = MODE == MODE_ROUND0 && c->r0_rank16; // round 0: 16 consecutive slots per thread
    const bool normal16 = MODE == MODE_NORMAL && c->r0_rank16;
    if (normal16)
        k2_lrank16_p1<<<xcd_grid(tiles, nb), RT_THREADS, 0, c->stream>>>(sorted, cntArr, B.state, want, B.tileagg, B.S,
                                                                         B.TPB, tiles, nbarg(nb, 4));
    else if (serial16)
        k2_r0rank_p1<<<xcd_grid(tiles, nb), RT_THREADS, 0, c->stream>>>(sorted, cntArr, B.state, want, B.tileagg, B.S,
                                                                        B.TPB, tiles, nbarg(nb, 4), B.huse, r0runs);
    else
        k2_lrerank_p1<MODE><<<xcd_grid(tiles, nb), RT_THREADS, 0, c->stream>>>(sorted, cntArr, B.state, want,
                                                                               B.tileagg, B.S, B.TPB, tiles, nbarg(nb, 4),
                                                                               B.huse, r0runs);
    KCHECK();
    k2_lrerank_p2<<<nb, 256, 0, c->stream>>>(cntArr, B.state, want, B.tileagg, B.mLn, B.leftg,
                                             MODE == MODE_FINAL ? 1 : 0, B.TPB);
    KCHECK();
    const u32 nsw = (MODE == MODE_ROUND0) ? c->r0_sweeps : 1u;
    if (normal16)
        k2_lrank16_p3<<<xcd_grid(tiles, nb), RT_THREADS, 0, c->stream>>>(sorted, lout, B.slA, B.slB, B.par, cntArr,
                                                                         B.state, want, B.tileagg, B.rank, B.mSn,
                                                                         B.nheads, B.S, B.TPB, tiles, nbarg(nb, 4));
    else if (serial16)
        k2_r0rank_p3<<<xcd_grid(tiles * nsw, nb), RT_THREADS, 0, c->stream>>>(
            sorted, lout, B.slA, B.slB, B.par, cntArr, B.state, want, B.tileagg, B.rank, B.n, B.mSn, B.nheads, B.S,
            B.TPB, tiles, nbarg(nb, 4), nsw, B.huse, r0runs);
    else
        k2_lrerank_p3<MODE><<<xcd_grid(tiles * nsw, nb), RT_THREADS, 0, c->stream>>>(
            sorted, lout, B.slA, B.slB, B.par, cntArr, B.state, want, B.tileagg, B.rank, B.T, B.bwt, B.n,
            B.mSn, B.nheads, B.S, B.TPB, tiles, nbarg(nb, 4), nsw, B.huse, r0runs);
    KCHECK();
}

// maxn = upper bound of the block lengths in the batch
void run_bwt_n(bz2hip_ctx *c, int nb, u32 maxn) {
    Batch &B = c->batch;
    if (nb == 0 || maxn == 0) return;
    const auto t_begin = std::chrono::steady_clock::now();
    const u32 tiles = (maxn + RTILE - 1) / RTILE;
    const int cb = (nb + 255) / 256;
    HIPCHK(hipMemsetAsync(B.ctl, 0, CTL_WORDS * sizeof(u32), c->stream));
    k2_reset<<<cb, 256, 0, c->stream>>>(B.n, B.present, B.mS, B.mL, B.mSn, B.mLn, B.hcur, B.huse,
                                        B.state, B.par, B.rounds, B.leftg, B.periodic, B.nheads, B.ctl, nb, c->r0bits);
    KCHECK();
    const int r0p = radix_passes(c);   // passes over an L-list: the whole 40-bit field
    const int r0p0 = radix_passes0(c); // passes of round 0: its r0bits key bits
    // The first pass reads the block bytes, so the keys are first written by its scatter.  The
    // sorted keys must end in keyB (the L-list is rebuilt into keyA): with an odd pass count the
    // passes go (T) -> keyB -> keyA -> ... -> keyB, so the "input" of pass 0 is nominally keyA.
    u64 *const A0 = (r0p0 & 1) ? B.keyA : B.keyB, *const B0 = (r0p0 & 1) ? B.keyB : B.keyA;
    // all passes but the last
    radix_sort5(c, nb, maxn, B.n, A0, B0, true, r0p0 - 1, 0, /*more_follow=*/true);
    u64 *const part = ((r0p0 - 1) & 1) ? B0 : A0; // sorted by all key bits but the top digit
    // Repeat distances are voted for BEFORE the last pass, on the list sorted by the low key bits:
    // neighbours with equal (whole) keys are still neighbours there unless an element that differs
    // in the top digit only sits between them, and the votes are a heuristic anyway (the rules are
    // exact whatever the votes: a poor candidate costs time, never correctness).  The chain votes ->
    // picks -> regions reads that list and T only: it runs on the other stream beside the last pass,
    // and its regions are ready for the re-rank of round 0, which finishes the groups they order
    // (round0_run_rule), and for every doubling round.  Voting again later changed no family's round
    // count and cost 0.3 ms (measured), so the regions are listed once.
    HIPCHK(hipEventRecord(c->ev_fork, c->stream));
    HIPCHK(hipStreamWaitEvent(c->stream_other, c->ev_fork, 0));
    {
        hipStream_t so = c->stream_other;
        HIPCHK(hipMemsetAsync(B.pvote, 0, (size_t)nb * VT * 2 * sizeof(u32), so));
        k2_vote0<<<xcd_grid((tiles + 3) / 4, nb), 64, 0, so>>>(part, B.n, B.huse, B.pvote, B.S, (tiles + 3) / 4, (u32)nb);
        KCHECK();
        k2_pick<<<nb, 256, 0, so>>>(B.pvote, B.n, B.cand, B.runs, nb);
        KCHECK();
        k2_viol<<<dim3(tiles, NCAND, nb), 256, 0, so>>>(B.T, B.n, B.cand, B.vagg, B.S, B.TPB);
        KCHECK();
        k2_runs<<<cb, 256, 0, so>>>(B.T, B.n, B.cand, B.vagg, B.runs, B.S, B.TPB, nb);
        KCHECK();
        HIPCHK(hipEventRecord(c->ev_join, so));
    }
    // the last pass: `part` -> keyB (the sorted keys must end in keyB: the L-list is rebuilt into keyA)
    radix_sort5(c, nb, maxn, B.n, part, part == B.keyA ? B.keyB : B.keyA, r0p0 == 1, r0p0, r0p0 - 1, false,
                /*dig_ready=*/r0p0 > 1);
    HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
    lrerank<MODE_ROUND0>(c, nb, maxn, B.n, ST_ACTIVE, B.keyB, B.keyA);
    // Round loop: k2_ctl of round r leaves its words in the control block (the round's own words are
    // double-buffered, each round zeroes its successor's: no memset in the loop), one small copy brings
    // them to a page-locked slot, and the host sizes the round's grids from them.  (Queueing round r
    // from the words of round r - 1 without waiting was measured: the loop is bound by the host's
    // launch rate in its tail, not by this wait, and the extra round of empty kernels at the end cost
    // more than the waits it saved: 25.05 -> 25.45 ms.)
    u32 finS = 0, finL = 0;
    bool anyfin = false;
    for (u32 r = 0;; r++) {
        k2_ctl<<<nb, 256, 0, c->stream>>>(B.n, B.mS, B.mL, B.mSn, B.mLn, B.hcur, B.huse, B.state, B.par,
                                          B.rounds, B.nheads, B.T, B.S, B.ctl, B.ctl + 16 + 16 * (r & 1),
                                          B.ctl + 16 + 16 * ((r + 1) & 1), nb);
        KCHECK();
        u32 *hc = c->h_ctl;
        HIPCHK(hipMemcpyAsync(hc, B.ctl, CTL_WORDS * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        const u32 *hr = hc + 16 + 16 * (r & 1);
        if (r == 0) {
            if (c->profile) {
                // pass 0 reads the block bytes (1 B) instead of 8-byte keys; with digit arrays every pass
                // but the last writes 1 B more per element and every later histogram reads 1 B, not 8
                const bool dg = c->radix_dig && c->radix_bits == 8;
                c->prof_bytes[P_RADIX_SCATTER] += ((r0p0 - 1) * 16.0 + 9.0 + (dg ? r0p0 - 1 : 0)) * (double)hc[8];
                c->prof_bytes[P_RADIX_HIST] += ((r0p0 - 1) * (dg ? 1.0 : 8.0) + 1.0) * (double)hc[8];
            }
            c->st_sorted_elems += (u64)hc[8] * r0p0;
        }
        anyfin = hc[4] != 0;
        finS = hc[5];
        finL = hc[6];
        if (!hr[RC_ANY]) break;
        const u32 maxS = hr[RC_MAXS], maxL = hr[RC_MAXL];
        if (c->debug >= 2)
            fprintf(stderr, "[bz2hip] round %u: maxS=%u maxL=%u sumS=%u sumL=%u\n", r, maxS, maxL, hr[RC_SUMS], hr[RC_SUML]);
        const bool small_l = maxL && maxL <= RTILE; // every L-list fits one tile: one kernel does the L-path
        c->st_s_elems += hr[RC_SUMS];
        if (!small_l) c->st_sorted_elems += (u64)hr[RC_SUML] * r0p;
        if (c->profile) {
            if (!small_l) {
                const bool dg = c->radix_dig && c->radix_bits == 8;
                c->prof_bytes[P_RADIX_SCATTER] += (r0p * 16.0 + (dg ? r0p - 1 : 0)) * (double)hr[RC_SUML];
                c->prof_bytes[P_RADIX_HIST] += (dg ? 8.0 + (r0p - 1) : r0p * 8.0) * (double)hr[RC_SUML];
            }
            // keys: read E 8 + gather rank 4 + write E 8; ssort: read E 8 + write E 8 (LDS sort)
            c->prof_bytes[P_BWT_KEYS] += 20.0 * ((double)hr[RC_SUML] + (double)hr[RC_SUMS]);
            c->prof_bytes[P_BWT_SSORT] += 16.0 * (double)hr[RC_SUMS];
        }
        {
            ProfScope ps(c, P_BWT_KEYS);
            const u32 mx = std::max(maxS, maxL);
            k2_keys<<<xcd_grid(2 * ((mx + RTILE - 1) / RTILE), nb), RT_THREADS, 0, c->stream>>>(
                B.slA, B.slB, B.keyA, B.par, B.mS, B.mL, B.state, B.huse, B.n, B.rank, B.S, (mx + RTILE - 1) / RTILE, nbarg(nb, 1));
            KCHECK();
        }
        // The S-sort and the L-path of a round touch disjoint elements (both only append to the
        // next S-list through the atomic counter mSn): when both exist the L-path runs on the
        // other stream.
        const bool fork = maxS && maxL && !small_l;
        if (fork) {
            HIPCHK(hipEventRecord(c->ev_fork, c->stream));
            HIPCHK(hipStreamWaitEvent(c->stream_other, c->ev_fork, 0));
        }
        if (maxS) {
            ProfScope ps(c, P_BWT_SSORT);
            k2_ssort<MODE_NORMAL><<<xcd_grid((maxS + SC - 1) / SC, nb), RT_THREADS, 0, c->stream>>>(
                B.slA, B.slB, B.slA, B.slB, B.par, B.mS, B.state, ST_ACTIVE, B.rank, B.T, B.bwt, B.n,
                B.mSn, B.leftg, B.huse, (u32 *)nullptr,
                B.runs, B.nheads, B.S, (maxS + SC - 1) / SC, nbarg(nb, 2));
            KCHECK();
            KCHECK();
        }
        if (small_l) {
            ProfScope ps(c, P_BWT_RERANK);
            k2_lsmall<<<nb, RT_THREADS, 0, c->stream>>>(B.keyA, B.slA, B.slB, B.par, B.mL, B.state, B.rank, B.T, B.bwt,
                                                        B.n, B.mSn, B.mLn, B.huse, B.runs, B.nheads, B.S);
            KCHECK();
        } else if (maxL) {
            if (fork) std::swap(c->stream, c->stream_other); // the drivers below launch on c->stream
            {
                // long-repeat rules on the L-list (after k2_keys, before the sort)
                ProfScope ps(c, P_BWT_KEYS);
                const u32 lt = (maxL + RTILE - 1) / RTILE;
                const u32 *runs = B.runs;
                k2_lap_p1<<<xcd_grid(lt, nb), RT_THREADS, 0, c->stream>>>(B.keyA, B.mL, B.state, B.huse, runs,
                                                                         (u32 *)nullptr, B.tileagg, B.S, B.TPB, lt, nbarg(nb, 4));
                KCHECK();
                k2_lap_p2<<<nb, 256, (size_t)B.TPB * 8 * sizeof(u32), c->stream>>>(B.mL, B.state, B.tileagg,
                                                                                   B.r2agg, B.TPB);
                KCHECK();
                k2_lap_p3<<<xcd_grid(lt, nb), RT_THREADS, 0, c->stream>>>(B.keyA, B.mL, B.state, B.huse, runs,
                                                                         B.tileagg, B.r2agg, B.S, B.TPB, lt,
                                                                         nbarg(nb, 4));
                KCHECK();
            }
            radix_sort5(c, nb, maxL, B.mL, B.keyA, B.keyB);
            // an even number of passes leaves the sorted list in keyA: the two buffers trade names
            if (!(r0p & 1)) std::swap(B.keyA, B.keyB);
            lrerank<MODE_NORMAL>(c, nb, maxL, B.mL, ST_ACTIVE, B.keyB, B.keyA);
            if (fork) {
                HIPCHK(hipEventRecord(c->ev_join, c->stream));
                std::swap(c->stream, c->stream_other);
                HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
            }
        }
    }
    {
        // every finished rotation's BWT byte, in one streaming pass (before the tie flush, which
        // overwrites the slots of tied rotations with the same bytes, one per slot)
        ProfScope ps(c, P_BWT_EMIT);
        k2_emit_bwt<<<xcd_grid(tiles, nb), RT_THREADS, 0, c->stream>>>(B.n, B.rank, B.T, B.bwt, B.S, tiles, (u32)nb);
        KCHECK();
    }
    // the loop above ended on a device round trip: only the emit pass and the tie flush remain
    c->st_bwt_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    if (anyfin) {
        // periodic blocks: flush the k-fold ties in list order
        if (finS) {
            ProfScope ps(c, P_BWT_SSORT);
            k2_ssort<MODE_FINAL><<<xcd_grid((finS + SC - 1) / SC, nb), RT_THREADS, 0, c->stream>>>(
                B.slA, B.slB, B.slA, B.slB, B.par, B.mS, B.state, ST_FIN, B.rank, B.T, B.bwt, B.n,
                B.mSn, B.leftg, B.huse, (u32 *)nullptr, (const u32 *)nullptr, B.nheads, B.S, (finS + SC - 1) / SC,
                nbarg(nb, 2));
            KCHECK();
        }
        if (finL) lrerank<MODE_FINAL>(c, nb, finL, B.mL, ST_FIN, B.keyA, B.keyB);
    }
    {
        ProfScope ps(c, P_BWT_EMIT);
        k2_finish<<<cb, 256, 0, c->stream>>>(B.n, B.rank, B.mS, B.mL, B.state, B.leftg, B.T, B.bwt,
                                             B.periodic, B.orig, B.S, nb, c->opts.tie_rule);
        KCHECK();
    }
    // periodic blocks: the reference's origin pointer depends on its sorter's tie order (k2_tie.hip)
    B.tie_items.clear();
    B.tie_running = false;
    if (anyfin && c->opts.tie_rule == BZ2HIP_TIE_REFERENCE) run_tie_launch(c, nb);
}
