// k1_rle1.hip -- block splitting, RLE1 and block CRC on the device.
//
// Replaces BZip2BlockCompressor.write / writeRun (S/BZip2BlockCompressor.swift:151-252) and
// BZip2CRC32 (S/BZip2CRC32.swift:32-105) for a window of input bytes resident in HBM.
//
// Acceptance rule (:198-227): a byte is refused when the bytes *flushed* so far exceed
// blockSize-6.  Runs are flushed when a different byte is accepted, or at 255 bytes.  With
//   head(i)  = i == 0 || in[i] != in[i-1]
//   rs(i)    = start of the maximal run containing i            (max-scan of head positions)
//   c(i)     = 5 if (i - rs(i) + 1) % 255 == 0                  (a full 255-chunk flushed)
//            + rem((i - rs(i-1)) % 255) if head(i), i > 0       (remainder of the run ending at i-1)
//   P(i)     = sum_{j<i} c(j)                                   (bytes flushed when i is offered,
//                                                                for a block that began at 0)
// a block that starts at s (possibly inside a run that ends at re0) has flushed
//   5*floor((i-s)/255)                         for i <= re0
//   enc(re0-s) + P(i) - P(re0+1)               for i >  re0
// so the first refused position is found by a monotone search on P.  Blocks form a serial
// chain (each start is the previous end); one wavefront walks it with 64-ary searches.
#include "ctx.hpp"

#define K1_TILE 4096
#define K1_NONE 0u

__device__ __forceinline__ u32 enc_rem(u32 r) { return r < 4 ? r : 5u; }

// phase 1: last head position + 1 per tile.  16 consecutive bytes per thread (one 16-byte load when
// the input is aligned and the 16 lie below n), the byte before them from the neighbouring lane.
__global__ __launch_bounds__(256) void k1_heads_p1(const u8 *__restrict__ in, u32 n,
                                                   u32 *__restrict__ agg, u32 tile0) {
    const u32 tile = tile0 + blockIdx.x;
    const u32 pos = tile * K1_TILE + threadIdx.x * 16u;
    __shared__ u32 sm[8];
    u32 w[4];
    if (pos + 16u <= n && (((uintptr_t)(in + pos)) & 15u) == 0) {
        const uint4 v = *(const uint4 *)(in + pos);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            u32 x = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u32 i = pos + q * 4 + j;
                if (i < n) x |= (u32)in[i] << (8 * j);
            }
            w[q] = x;
        }
    }
    u32 prv = __shfl_up(w[3] >> 24, 1, 64);
    if (lane_id() == 0 && pos > 0 && pos < n) prv = in[pos - 1];
    // bytes that differ from the byte before them: x = the 16 bytes xor the 16 shifted up by one
    u32 last = 0;
    u32 pb = prv;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const u32 sh = (w[q] << 8) | pb;
        pb = w[q] >> 24;
        u32 x = w[q] ^ sh; // byte j != 0: a head at pos + 4q + j
        if (pos + 4u * q + 4u > n) x &= (pos + 4u * q >= n) ? 0u : (0xffffffffu >> (8u * (pos + 4u * q + 4u - n)));
        if (pos == 0 && q == 0) x |= 1u; // position 0 is a head
        if (x) last = pos + 4u * q + (31u - (u32)__clz(x)) / 8u + 1u;
    }
    u32 t = block_reduce_max<256>(last, sm);
    if (threadIdx.x == 0) agg[tile] = t;
}

// exclusive scans over a flat array of tile aggregates, tiles [first, count), 1024 tiles per
// workgroup.  Out of place (raw -> out), so a workgroup takes its carry from a reduction of the
// raw values before its tiles (at most 64 coalesced loads per thread for 256 MiB of input) instead
// of waiting for the workgroups before it; input chunks are scanned as they arrive, each launch
// reducing from tile 0.  The workgroup that holds the last tile leaves the running total.
template <bool MAXOP>
__global__ __launch_bounds__(1024) void k1_scan_tiles(const u32 *__restrict__ raw, u32 *__restrict__ out,
                                                      u32 first, u32 count, u32 *__restrict__ total_out) {
    __shared__ u32 sm[20];
    const u32 t0 = first + blockIdx.x * 1024u;
    u32 acc = 0;
#pragma unroll 8
    for (u32 q = threadIdx.x; q < t0; q += 1024u) {
        const u32 v = raw[q];
        acc = MAXOP ? max(acc, v) : acc + v;
    }
    const u32 carry = MAXOP ? block_reduce_max<1024>(acc, sm) : block_reduce_add<1024>(acc, sm);
    __syncthreads();
    const u32 t = t0 + threadIdx.x;
    const u32 v = (t < count) ? raw[t] : 0;
    u32 tot, ex;
    if (MAXOP) {
        const u32 inc = block_incl_scan_max<1024>(v, sm, &tot);
        const u32 prev = __shfl_up(inc, 1, 64);
        if (lane_id() == 0) {
            ex = 0;
            for (int q = 0; q < wave_id(); q++) ex = max(ex, sm[q]);
        } else {
            ex = prev;
        }
        ex = max(ex, carry);
        tot = max(tot, carry);
    } else {
        ex = carry + block_excl_scan_add<1024>(v, sm, &tot);
        tot += carry;
    }
    if (t < count) out[t] = ex;
    if (threadIdx.x == 0 && total_out && t0 + 1024u >= count) *total_out = tot;
}

// The flushed-length prefix P and the run starts are kept per CELL of 256 input bytes, not per
// byte: Hc[c] = last head (+1) before the cell, Pc[c] = P at the cell's first byte (tile-local;
// readers add the scanned tile sums agg_c).  8 bytes per 256 input bytes instead of 8 per byte; a
// reader that needs P inside a cell recomputes it from the cell's 256 bytes (cell_compute).
#define K1_CELL 256
#define K1_CPT (K1_TILE / K1_CELL) // cells per tile

// phase 3: cell records, the 16-bit prefix and tile sums of c(i).  Every thread owns 16 consecutive
// bytes (one 16-byte load) and walks them with the chunk phase L = (i - rs(i)) % 255 in a register
// (no division per byte); the run start that enters a thread comes from a max-scan of "last head"
// over the threads before it, the prefix from a scan of the threads' sums.
__global__ __launch_bounds__(256) void k1_cells(const u8 *__restrict__ in, u32 n,
                                                const u32 *__restrict__ agg_head,
                                                u32 *__restrict__ Hc, u32 *__restrict__ agg_c,
                                                u32 *__restrict__ Pc, u16 *__restrict__ P16, u32 tile0) {
    const u32 tile = tile0 + blockIdx.x;
    const u32 pos = tile * K1_TILE + threadIdx.x * 16u;
    const int lane = lane_id(), wv = wave_id();
    __shared__ u32 sm[8];
    __shared__ u32 red[8];
    u32 w[4];
    if (pos + 16u <= n && (((uintptr_t)(in + pos)) & 15u) == 0) {
        const uint4 v = *(const uint4 *)(in + pos);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            u32 x = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u32 i = pos + q * 4 + j;
                if (i < n) x |= (u32)in[i] << (8 * j);
            }
            w[q] = x;
        }
    }
    u32 prv = __shfl_up(w[3] >> 24, 1, 64);
    if (lane == 0 && pos > 0 && pos < n) prv = in[pos - 1];
    u32 heads = 0, lastp1 = 0;
    {
        u32 pb = prv;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const u32 i = pos + j;
            const u32 c = (w[j >> 2] >> (8 * (j & 3))) & 255u;
            if (i < n && (i == 0 || c != pb)) {
                heads |= 1u << j;
                lastp1 = i + 1;
            }
            pb = c;
        }
    }
    // last head (+1) before this thread
    const u32 inc = wave_incl_scan_max(lastp1);
    if (lane == 63) sm[wv] = inc;
    __syncthreads();
    u32 carry = agg_head[tile];
    for (int q = 0; q < wv; q++) carry = max(carry, sm[q]);
    const u32 up = __shfl_up(inc, 1, 64);
    if (lane > 0) carry = max(carry, up);
    // c(i) of the 16 bytes: 4 bits each (c <= 10)
    u32 L = carry ? (pos - (carry - 1u)) % 255u : 0u; // bytes of the entering run's current chunk
    u64 cpack = 0;
    u32 tot = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const u32 i = pos + j;
        u32 c = 0;
        if (i < n) {
            if ((heads >> j) & 1u) {
                if (i > 0) c = enc_rem(L); // remainder of the run that ended at i - 1
                L = 0;
            }
            if (++L == 255u) { // a full chunk is flushed with this byte
                c += 5u;
                L = 0;
            }
        }
        cpack |= (u64)c << (4 * j);
        tot += c;
    }
    u32 total;
    const u32 base = block_excl_scan_add<256>(tot, red, &total);
    if (threadIdx.x == 0) agg_c[tile] = total;
    if ((threadIdx.x & 15u) == 0 && pos < n) {
        Hc[pos / K1_CELL] = carry;
        Pc[pos / K1_CELL] = base;
    }
    // tile-local prefix per byte (< 1.25 x K1_TILE: 16 bits)
    u32 run = base;
    u32 o[8];
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
        const u32 p0 = run;
        run += (u32)(cpack >> (4 * j)) & 15u;
        const u32 p1 = run;
        run += (u32)(cpack >> (4 * j + 4)) & 15u;
        o[j >> 1] = p0 | (p1 << 16);
    }
    if (pos + 16u <= n) {
        uint4 *dst = (uint4 *)(P16 + pos);
        dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
        dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
    } else {
#pragma unroll
        for (int j = 0; j < 16; j++)
            if (pos + j < n) P16[pos + j] = (u16)(o[j >> 1] >> (16 * (j & 1)));
    }
}

// the cell that position n falls into when n is a multiple of the cell: Hc = last head of all
// (+1), P(n) = total
__global__ void k1_sentinels(u32 *Hc, u32 *Pc, u16 *P16, u32 *agg_c, const u32 *agg_head_total, u32 n,
                             const u32 *total) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const u32 t = n / K1_TILE;
        if (n % K1_TILE == 0) agg_c[t] = *total; // the tile n falls into does not exist yet
        P16[n] = (u16)(*total - agg_c[t]);        // P(n) = total
        if (n % K1_CELL == 0) {
            Hc[n / K1_CELL] = *agg_head_total;
            Pc[n / K1_CELL] = *total - agg_c[t];
        }
    }
}

// accessor of the searched prefix array
struct PrefixArr { // absolute flushed-length prefix = tile-local value (16 bits) + scanned tile sum
    const u16 *P;
    const u32 *agg;
    __device__ __forceinline__ u32 operator()(u32 i) const { return P[i] + agg[i / K1_TILE]; }
};

// first i in (lo, hi] with arr[i] > thr (arr non-decreasing); returns hi+1 if none.  One wave.
template <typename ARR>
__device__ u32 wave_search_gt(const ARR arr, u32 lo, u32 hi, u32 thr, bool top_known = false) {
    const int lane = lane_id();
    if (lo >= hi) return hi + 1;
    // invariant: answer in (lo, hi] or none; first check arr[hi] (unless the caller has)
    if (!top_known) {
        u32 top = arr(hi);
        if (!(top > thr)) return hi + 1;
    }
    // 256-ary steps: every lane probes 4 positions (independent loads), so a level costs one
    // memory latency and 46 M positions need 4 levels instead of 5 (this chain is pure latency)
    while (hi - lo > 1) {
        const u32 width = hi - lo;
        const u32 step = (width + 255) / 256;
        u32 best = 256; // first probe position (0..255) whose value exceeds thr
#pragma unroll
        for (int j = 0; j < 4; j++) {
            u32 p = (u32)lane * 4u + (u32)j;
            u32 idx = lo + (p + 1) * step;
            if (idx > hi || idx < lo) idx = hi;
            bool pred = arr(idx) > thr;
            u64 m = __ballot(pred);
            if (m) best = min(best, (u32)(__ffsll((unsigned long long)m) - 1) * 4u + (u32)j);
        }
        // best < 256 always: the last probe is clamped to hi and arr(hi) > thr
        u32 nhi = lo + (best + 1) * step;
        if (nhi > hi || nhi < lo) nhi = hi;
        u32 nlo = lo + best * step;
        lo = nlo;
        hi = nhi;
    }
    return hi;
}

// One probe level over (lo, hi] plus arr(lo) itself: narrows to (nlo, nhi] when arr(lo) <= thr <
// arr(hi) (the first i with arr[i] > thr is then inside); false when the range does not bracket it.
template <typename ARR>
__device__ bool wave_bracket(const ARR arr, u32 lo, u32 hi, u32 thr, u32 *nlo, u32 *nhi) {
    const int lane = lane_id();
    const u32 width = hi - lo;
    const u32 step = (width + 255) / 256;
    const u32 vlo = arr(lo);
    u32 best = 256;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        u32 p = (u32)lane * 4u + (u32)j;
        u32 idx = lo + (p + 1) * step;
        if (idx > hi || idx < lo) idx = hi;
        bool pred = arr(idx) > thr;
        u64 m = __ballot(pred);
        if (m) best = min(best, (u32)(__ffsll((unsigned long long)m) - 1) * 4u + (u32)j);
    }
    if (vlo > thr || best == 256) return false;
    u32 b = lo + (best + 1) * step;
    if (b > hi || b < lo) b = hi;
    *nlo = lo + best * step;
    *nhi = b;
    return true;
}

// ---------------------------------------------------------------- split hints
// Where a block ends depends on where it starts only through a threshold on P that lies within
// +-10 of P(s) + limit (header comment; the cut first run shifts it by at most two flushes).  So
// for every cell start g = 256 c, in parallel, before the chain:
//   hint[2c]     = last e in [g, n] with P(e) <= P(g) + limit - 16
//   hint[2c + 1] = a position shortly after it with P(e) > P(g) + limit + 16, or n
// and a block that starts in cell c ends in (hint[2c], hint[2(c+1) + 1]]: about 300 positions for
// run-free data, which the chain settles with one probe level instead of a search from scratch.
// The chain checks the bracket against its own threshold and falls back to the search when it
// does not hold, so the table is an accelerator only.
#define K1_HINT_SLACK 16u
__device__ __forceinline__ u32 hint_last_le(const u32 *__restrict__ Pc, const PrefixArr &P, u32 n, u32 c0,
                                            u32 c1, u32 T) {
    u32 lo = c0, hi = c1; // P(256 lo) <= T; the answer's cell is in [lo, hi]
    while (lo < hi) {
        const u32 mid = (lo + hi + 1u) >> 1;
        if (Pc[mid] + P.agg[mid / K1_CPT] <= T) lo = mid;
        else hi = mid - 1u;
    }
    u32 a = lo * K1_CELL, b = min(a + K1_CELL - 1u, n);
    while (a < b) {
        const u32 mid = (a + b + 1u) >> 1;
        if (P(mid) <= T) a = mid;
        else b = mid - 1u;
    }
    return a;
}
__global__ __launch_bounds__(256) void k1_hints(const u32 *__restrict__ Pc, const u16 *__restrict__ Ploc,
                                                const u32 *__restrict__ agg_c, u32 n, u32 block_size,
                                                u32 *__restrict__ hint) {
    // a wave takes 64 consecutive cells: their answers are consecutive too, so the whole wave first
    // finds the cell of lane 0's answer (64-ary, 3 levels for any block size) and every lane then
    // searches the 128 cells from there on
    const u32 lane = (u32)lane_id();
    const u32 cw = (blockIdx.x * 4u + (u32)wave_id()) * 64u;
    const u32 last = n / K1_CELL; // cells 0..last have a record (k1_cells, k1_sentinels)
    if (cw > last) return;
    const u32 c = min(cw + lane, last);
    const PrefixArr P{Ploc, agg_c};
    const u32 limit = block_size - 6u;
    // a block never swallows more than ~51x its size of input (k1_split)
    const u32 span = (u32)(((u64)block_size * 52ull + 1024ull) / K1_CELL + 2ull);
    const u32 c1 = (u32)min((u64)last, (u64)c + span);
    const u32 Tlo = Pc[c] + agg_c[c / K1_CPT] + limit - K1_HINT_SLACK;
    const u32 T0 = (u32)__builtin_amdgcn_readfirstlane((int)Tlo);
    u32 a0 = cw;
    {
        u32 lo = cw, hi = (u32)min((u64)last, (u64)cw + span); // P(256 lo) <= T0; answer cell in [lo, hi]
        while (lo < hi) {
            const u32 step = (hi - lo + 63u) / 64u;
            const u32 idx = min(lo + (lane + 1u) * step, hi);
            const u32 k = (u32)__popcll(__ballot(Pc[idx] + agg_c[idx / K1_CPT] <= T0)); // a prefix of the lanes
            const u32 nlo = min(lo + k * step, hi);
            hi = min(lo + (k + 1u) * step - 1u, hi);
            lo = nlo;
        }
        a0 = lo;
    }
    u32 lo = max(a0, c), hi = min(a0 + 128u, c1);
    if (hi < lo) hi = lo;
    if (hi < c1 && Pc[hi] + agg_c[hi / K1_CPT] <= Tlo) { // the window is too short (answers in a region of long runs)
        lo = hi;
        hi = c1;
    }
    const u32 e_lo = hint_last_le(Pc, P, n, lo, hi, Tlo);
    // upper end: any position whose P exceeds the upper threshold will do, as long as it is near
    const u32 Thi = Tlo + 2u * K1_HINT_SLACK;
    const u32 q1 = min(e_lo + 64u, n), q2 = min(e_lo + 256u, n);
    const u32 p1 = P(q1), p2 = P(q2);
    u32 e_hi;
    if (p1 > Thi || q1 == n) e_hi = q1;
    else if (p2 > Thi || q2 == n) e_hi = q2;
    else e_hi = min(hint_last_le(Pc, P, n, q2 / K1_CELL, max(c1, q2 / K1_CELL), Thi) + 1u, n);
    if (cw + lane <= last) {
        hint[2 * c] = e_lo;
        hint[2 * c + 1] = e_hi;
    }
}

// The first i in (lo, hi] with arr[i] > thr when hi - lo <= 512 and arr(lo) <= thr < arr(hi): one
// level of 512 probes.  false when the range does not bracket it.
template <typename ARR>
__device__ bool wave_bracket_exact(const ARR arr, u32 lo, u32 hi, u32 thr, u32 *e) {
    const u32 lane = (u32)lane_id();
    const u32 vlo = arr(lo);
    u32 best = 512;
#pragma unroll
    for (int j = 7; j >= 0; j--) {
        const u32 idx = min(lo + 1u + (u32)j * 64u + lane, hi);
        const u64 m = __ballot(arr(idx) > thr);
        if (m) best = (u32)j * 64u + (u32)(__ffsll((unsigned long long)m) - 1);
    }
    if (vlo > thr || best == 512) return false;
    *e = min(lo + 1u + best, hi);
    return true;
}

// ---------------------------------------------------------------- cells, read side (one wave)
struct CellArrs {
    const u8 *in;
    const u32 *Hc, *Pc, *agg;
    u32 n;
};
struct CellHeadArr { // last head (+1) before a cell
    const u32 *Hc;
    __device__ __forceinline__ u32 operator()(u32 c) const { return Hc[c]; }
};
struct CellRaw { // what cell_compute needs from memory: issued together, used after one latency
    u32 b[4];  // in[c*256 + r*64 + lane]
    u32 prev0; // in[c*256 - 1]
    u32 H, P0;
};
__device__ __forceinline__ void cell_load(const CellArrs &A, u32 c, CellRaw &R) {
    const u32 base = c * K1_CELL;
    const int lane = lane_id();
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const u32 i = base + r * 64 + lane;
        R.b[r] = A.in[i < A.n ? i : A.n - 1];
    }
    R.prev0 = A.in[base ? min(base - 1, A.n - 1) : 0];
    R.H = A.Hc[c];
    R.P0 = A.Pc[c] + A.agg[c / K1_CPT];
}
// pv[r] = P(c*256 + r*64 + lane) (meaningful for positions <= n); hm[r] = heads of row r.
// The arithmetic is k1_cells' (same c(i)), started from the cell record.
__device__ __forceinline__ void cell_compute(const CellArrs &A, u32 c, const CellRaw &R, u32 (&pv)[4],
                                             u64 (&hm)[4]) {
    const int lane = lane_id();
    const u32 base = c * K1_CELL;
    u32 carry = R.H, run = R.P0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const u32 rowbase = base + r * 64;
        const u32 i = rowbase + lane;
        u32 pb = __shfl_up(R.b[r], 1, 64);
        const u32 edge = r == 0 ? R.prev0 : (u32)__builtin_amdgcn_readlane((int)R.b[r > 0 ? r - 1 : 0], 63);
        if (lane == 0) pb = edge;
        const bool head = i < A.n && (i == 0 || R.b[r] != pb);
        const u64 m = __ballot(head);
        hm[r] = m;
        const u64 below = m & ((1ull << lane) - 1ull);
        const u32 prevp1 = below ? (rowbase + (63 - __clzll((unsigned long long)below)) + 1) : carry;
        u32 cval = 0;
        if (i < A.n) {
            const u32 rsi = head ? i : (prevp1 - 1);
            if ((i - rsi + 1) % 255u == 0) cval = 5;
            if (head && i > 0) cval += enc_rem((i - (prevp1 - 1)) % 255u);
        }
        const u32 inc = wave_incl_scan_add(cval);
        pv[r] = run + inc - cval;
        run += (u32)__builtin_amdgcn_readlane((int)inc, 63);
        if (m) carry = rowbase + (63 - __clzll((unsigned long long)m)) + 1;
    }
}
// first head position > s inside cell c (heads of computed rows); ~0 if none
__device__ __forceinline__ u32 cell_first_head_after(const u64 (&hm)[4], u32 c, u32 s) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const u32 rowbase = c * K1_CELL + r * 64;
        u64 m = hm[r];
        if (rowbase <= s) { // drop the positions <= s
            const u32 cut = s - rowbase + 1;
            m = cut >= 64 ? 0ull : (m >> cut) << cut;
        }
        if (m) return rowbase + (u32)(__ffsll((unsigned long long)m) - 1);
    }
    return 0xffffffffu;
}

// serial chain over blocks; one wave.  starts[k] for k <= count (starts[count] = end of last).
// The chain is pure memory latency (each block's start is the previous block's end), so a block
// costs as few dependent loads as possible: the run-end probe also fetches the prefix it will need,
// and the search first tries a narrow bracket around "as long as the previous block".
// The chain begins at s0 (a block start; a run that straddles s0 is cut there, so nothing before
// s0 matters) and takes blocks whose start lies below `stop` (sharding: the next shard continues
// from starts[count]).
__global__ __launch_bounds__(64) void k1_split(const u8 *__restrict__ in, const u32 *__restrict__ Hc,
                                               const u32 *__restrict__ Pc,
                                               const u16 *__restrict__ Ploc,
                                               const u32 *__restrict__ agg_c, u32 n, u32 block_size,
                                               u32 max_blocks, u32 *__restrict__ starts,
                                               u32 *__restrict__ binfo,
                                               u32 *__restrict__ count_out, u32 s0, u32 stop,
                                               const u32 *__restrict__ hint) {
    const u32 limit = block_size - 6;
    const PrefixArr P{Ploc, agg_c};
    const CellArrs A{in, Hc, Pc, agg_c, n};
    const u32 cmax = n / K1_CELL;
    u32 s = s0, cnt = 0, plen = 0, perr = 0;
    while (s < n && s < stop && cnt < max_blocks) {
        if (lane_id() == 0) starts[cnt] = s;
        // end of the run containing s: almost always within the next 64 bytes; the same step
        // fetches P(idx + 1), which is P(re0 + 1) for the lane that finds the run end
        u32 re0, p_re0p1 = 0, h_lo = 0, h_hi = 0;
        bool have_p = false;
        {
            const u32 idx = s + 1u + (u32)lane_id();
            if (hint) { // the bracket of this block's end, requested with the run-end probe
                const u32 hc = s / K1_CELL;
                h_lo = hint[2 * hc];
                h_hi = hc < cmax ? hint[2 * hc + 3] : n;
            }
            // a head at idx (the end of the data counts as one)
            const u32 hb = in[min(idx, n - 1)], hbp = in[min(idx - 1u, n - 1)];
            const bool pred = idx <= n && (idx == n || hb != hbp);
            const u32 pnext = (idx + 1u <= n) ? P(idx + 1u) : 0u;
            const u64 m = __ballot(pred);
            if (m) {
                const int l0 = __ffsll((unsigned long long)m) - 1;
                re0 = s + 1u + (u32)l0;
                p_re0p1 = __shfl(pnext, l0, 64);
                have_p = re0 + 1u <= n;
            } else {
                // a long run: the first cell whose "last head before" lies beyond s follows the
                // cell that holds the run's end
                const u32 c0 = (s + 64u) / K1_CELL;
                const u32 c1 = wave_search_gt(CellHeadArr{Hc}, c0, cmax, s + 1u);
                const u32 c = c1 - 1; // c1 = cmax + 1 when no cell start sees a head beyond s: the last cell
                CellRaw R;
                u32 pv[4];
                u64 hm[4];
                cell_load(A, c, R);
                cell_compute(A, c, R, pv, hm);
                re0 = cell_first_head_after(hm, c, s);
                if (re0 == 0xffffffffu) re0 = n; // the run reaches the end of the data
            }
        }
        if (re0 > n) re0 = n;
        u32 e, pre = 0;
        u32 cand = s + 255u * (limit / 5u + 1u);
        if (cand <= re0 && cand > s) {
            e = cand;
        } else if (re0 >= n) {
            e = n;
        } else {
            u32 L0 = re0 - s;
            u32 enc0 = 5u * (L0 / 255u) + enc_rem(L0 % 255u);
            u32 hi = n;
            // a block never swallows more than ~51x its size of input
            u64 cap = (u64)s + (u64)block_size * 52ull + 1024ull;
            if (cap < (u64)hi) hi = (u32)cap;
            if (enc0 > limit) {
                e = re0 + 1; // the first run alone overfills: refused right after the next head
            } else {
                pre = have_p ? p_re0p1 : P(re0 + 1);
                const u32 thr = limit - enc0 + pre;
                // blocks of one kind of data are about equally long.  When the last block was as long as
                // the one before (run-free data: always bs - 4 bytes), one probe of the 64 positions
                // around "as long again" settles this block in a single dependent load.
                bool done = false, br = false;
                u32 nlo = 0, nhi = 0;
                if (hint) {
                    const u32 lo2 = max(re0, h_lo), hi2 = min(hi, h_hi);
                    if (lo2 < hi2) {
                        if (hi2 - lo2 <= 512u) done = wave_bracket_exact(P, lo2, hi2, thr, &e);
                        else br = wave_bracket(P, lo2, hi2, thr, &nlo, &nhi);
                    }
                }
                if (!done && !br && plen && perr == 0) {
                    const u32 c = s + plen;
                    if (c > re0 + 33u && (u64)c + 31ull <= (u64)hi) {
                        const u32 idx = c - 32u + (u32)lane_id();
                        const u64 m = __ballot(P(idx) > thr);
                        if (m && !(m & 1ull)) { // P(c - 32) <= thr < P(some probe): the first such probe
                            e = c - 32u + (u32)(__ffsll((unsigned long long)m) - 1);
                            done = true;
                        }
                    }
                }
                // otherwise bracket around the last length
                if (!done && !br && plen) {
                    // as wide as the last prediction was wrong (x4), at least 1.5 % of the length
                    const u32 half = min(max(2048u, 4u * perr + plen / 64u), max(2048u, plen / 2u));
                    const u32 c = s + plen;
                    const u32 lo2 = max(re0, c > half ? c - half : 0u);
                    const u32 hi2 = (u32)min((u64)hi, (u64)c + half);
                    if (lo2 < hi2) br = wave_bracket(P, lo2, hi2, thr, &nlo, &nhi);
                }
                if (!done) {
                    e = br ? wave_search_gt(P, nlo, nhi, thr, true) : wave_search_gt(P, re0, hi, thr);
                    if (e > hi) e = (hi == n) ? n : hi; // not expected when hi < n
                }
            }
        }
        // for the emit: where the block's first run ends and P just after it (k1_emit)
        if (lane_id() == 0) {
            binfo[2 * cnt] = re0;
            binfo[2 * cnt + 1] = pre;
        }
        cnt++;
        perr = (e - s > plen) ? (e - s - plen) : (plen - (e - s));
        plen = e - s;
        s = e;
    }
    if (lane_id() == 0) {
        starts[cnt] = s;
        *count_out = cnt;
    }
}

// ---------------------------------------------------------------- per-block RLE1 emit
// Tile x of block b is the global tile (starts[b] / K1_TILE + x) cut to [s, e): the tiles of the
// scan and of the emit coincide, so a tile's run starts follow from its own bytes and the scan's
// "last head before this tile" (agg_head) -- no per-byte array is read.  The block start is a
// forced head (a run that straddles s is cut there, S/BZip2BlockCompressor.swift:198-227).
// 3-phase scan of emitted bytes.  Every thread owns 16 consecutive bytes (one 16-byte load) and
// walks them serially with the run start in a register; the run start that enters a thread comes
// from a max-scan of "last head" over the threads before it.
struct EmitLane {
    u32 w[4];      // the 16 bytes
    u32 heads;     // bit j: byte j starts a run (inside [s, e))
    u32 lit;       // bit j: byte j is emitted as itself (the first four of a 255-chunk)
    u32 cnt;       // bit j: a count byte follows byte j (its chunk ends there with >= 4 bytes)
    u32 kprev;     // chunk phase of the byte before the 16: (its distance from the run start) mod 255
};

// Loads, head detection and the emit decisions of the thread's 16 bytes as bit masks; returns the
// thread's emitted-byte count.  EDGE = the tile holds the block's first or last byte (bounds are
// tested per byte); an inner tile lies wholly inside (s, e) with at least one byte of the block
// behind it, so every test against s and e is dropped.
template <bool EDGE>
__device__ __forceinline__ u32 emit_lane(const u8 *__restrict__ in, const u32 *__restrict__ agg_head,
                                         u32 s, u32 e, u32 gt, u32 *sm /* 8 words */, EmitLane &L) {
    const int lane = lane_id(), wv = wave_id();
    const u32 pos = gt * K1_TILE + threadIdx.x * 16u;
    // the 16 bytes: one aligned 16-byte load when they all lie below e (bytes before s belong to
    // the window too), single bytes at the block's end and for an unaligned input
    if ((!EDGE || pos + 16u <= e) && (((uintptr_t)(in + pos)) & 15u) == 0) {
        const uint4 v = *(const uint4 *)(in + pos);
        L.w[0] = v.x; L.w[1] = v.y; L.w[2] = v.z; L.w[3] = v.w;
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            u32 x = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u32 i = pos + q * 4 + j;
                if (!EDGE || i < e) x |= (u32)in[i] << (8 * j);
            }
            L.w[q] = x;
        }
    }
    // neighbours: last byte of the thread before, first byte of the thread after
    u32 prv = __shfl_up(L.w[3] >> 24, 1, 64);
    u32 nxt = __shfl_down(L.w[0] & 255u, 1, 64);
    if (lane == 0 && (!EDGE || (pos > s && pos < e))) prv = in[pos - 1];
    if (lane == 63 && (!EDGE || pos + 16u < e)) nxt = in[pos + 16u];
    u32 heads = 0, lastp1 = 0, valid = 0xffffu;
    u32 pb = prv;
    if (EDGE) valid = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const u32 i = pos + j;
        const u32 c = (L.w[j >> 2] >> (8 * (j & 3))) & 255u;
        bool h;
        if (EDGE) {
            const bool in_blk = i >= s && i < e;
            valid |= in_blk ? 1u << j : 0u;
            h = in_blk && (i == s || c != pb);
        } else {
            h = c != pb;
        }
        if (h) {
            heads |= 1u << j;
            lastp1 = i + 1;
        }
        pb = c;
    }
    L.heads = heads;
    // last head (+1) before this thread: the tile's carry, the block start, the threads before
    const u32 inc = wave_incl_scan_max(lastp1);
    if (lane == 63) sm[wv] = inc;
    __syncthreads();
    u32 carry = max(agg_head[gt], s + 1u); // (a run that straddles s is cut there: s is a head whatever the bytes say)
    for (int q = 0; q < wv; q++) carry = max(carry, sm[q]);
    const u32 up = __shfl_up(inc, 1, 64);
    if (lane > 0) carry = max(carry, up);
    const u32 r0 = carry - 1u; // start of the run that enters byte 0
    // a chunk ends at byte j when the next byte differs or lies beyond the block (or at 255 bytes)
    const bool d15 = (EDGE && pos + 16u >= e) || nxt != (L.w[3] >> 24);
    const u32 brk = (((heads | ~valid) >> 1) & 0x7fffu) | (d15 ? 0x8000u : 0u);
    u32 k = pos >= r0 ? (pos - r0 + 254u) % 255u : 0u;
    L.kprev = k;
    u32 lit = 0, cnt = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        k = (k == 254u) ? 0u : k + 1u;
        k = ((heads >> j) & 1u) ? 0u : k;
        const bool ends = ((brk >> j) & 1u) || k == 254u;
        lit |= (k < 4u) ? 1u << j : 0u;
        cnt |= (ends && k >= 3u) ? 1u << j : 0u;
    }
    L.lit = lit & valid;
    L.cnt = cnt & valid;
    return (u32)__popc(L.lit) + (u32)__popc(L.cnt);
}

#define K1_EMIT_STAGE (K1_TILE + K1_TILE / 4 + 64) // a run of 4 emits 5 bytes: at most 1.25 per input byte
// Bytes the block has emitted before the tile's first byte a, without a pass over the tiles before:
// inside the block's first run (a <= re0) that is 5 per full 255-chunk plus the open chunk; after
// it, enc(first run) + P(a) - P(re0 + 1) -- the flushed-length prefix the scan already holds at
// tile starts (agg_c) and the split recorded per block (binfo) -- plus the open chunk of the run
// that enters the tile.  An open chunk of k bytes has emitted min(k, 4) bytes, and its count byte
// too once the run has ended (a is a head).
__device__ __forceinline__ u32 emit_base(const u8 *__restrict__ in, const u32 *__restrict__ agg_head,
                                         const u32 *__restrict__ agg_c, const u32 *__restrict__ binfo,
                                         u32 b, u32 s, u32 gt) {
    const u32 a = gt * K1_TILE;
    if (a <= s) return 0;
    const u32 re0 = binfo[2 * b], pre = binfo[2 * b + 1];
    const bool head = in[a] != in[a - 1];
    if (a <= re0) {
        const u32 L = a - s, k = L % 255u;
        return 5u * (L / 255u) + (head ? enc_rem(k) : min(k, 4u));
    }
    const u32 L0 = re0 - s;
    const u32 enc0 = 5u * (L0 / 255u) + enc_rem(L0 % 255u);
    const u32 k = (a - (agg_head[gt] - 1u)) % 255u;
    return enc0 + (agg_c[gt] - pre) + (head ? enc_rem(k) : min(k, 4u));
}

template <bool EDGE>
__device__ __forceinline__ void emit_tile(const u8 *__restrict__ in, const u32 *__restrict__ agg_head,
                                          u32 s, u32 e, u32 gt, u32 ebase, u32 b, u8 *__restrict__ T,
                                          u32 *__restrict__ present, u32 *__restrict__ nOut, u32 S) {
    __shared__ u32 sm[8];
    __shared__ u32 red[8];
    __shared__ u32 pres[8 * 8]; // the bitmap's 8 words, each 8 times (lane mod 8): atomics of one wave on one word are served one by one
    __shared__ __attribute__((aligned(16))) u8 stage[K1_EMIT_STAGE];
    if (threadIdx.x < 64) pres[threadIdx.x] = 0;
    u32 *const mypres = pres + (threadIdx.x & 7u);
    EmitLane L;
    const u32 tot = emit_lane<EDGE>(in, agg_head, s, e, gt, sm, L);
    u32 total;
    u32 o = block_excl_scan_add<256>(tot, red, &total); // has a barrier after pres is cleared
    // the thread's bytes into the staging tile (a value enters the bitmap once per run: at its head)
    u32 k = L.kprev;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        k = (k == 254u) ? 0u : k + 1u;
        k = ((L.heads >> j) & 1u) ? 0u : k;
        const u32 c = (L.w[j >> 2] >> (8 * (j & 3))) & 255u;
        if ((L.lit >> j) & 1u) {
            stage[o++] = (u8)c;
            if ((L.heads >> j) & 1u) atomicOr(&mypres[(c >> 5) * 8u], 1u << (c & 31));
        }
        if ((L.cnt >> j) & 1u) {
            const u32 cb = k - 3u;
            stage[o++] = (u8)cb;
            atomicOr(&mypres[(cb >> 5) * 8u], 1u << (cb & 31));
        }
    }
    __syncthreads();
    u8 *out = T + (size_t)b * S + ebase;
    // copy out: single bytes up to the first 4-byte boundary of the destination and after the
    // last, whole words in between (two aligned LDS words funnel-shifted per destination word)
    {
        const u32 mis = (u32)((4u - ((uintptr_t)out & 3u)) & 3u); // bytes before the first boundary
        const u32 headn = min(mis, total);
        if (threadIdx.x < headn) out[threadIdx.x] = stage[threadIdx.x];
        const u32 nwords = (total - headn) / 4u;
        u32 *outw = (u32 *)(out + headn);
        const u32 *st32 = (const u32 *)stage; // stage is 16-byte aligned
        const u32 sh = headn & 3u;            // stage offset of destination word d is headn + 4 d
        for (u32 d = threadIdx.x; d < nwords; d += 256) {
            const u32 lo = st32[d + (headn >> 2)], hi = st32[d + (headn >> 2) + 1];
            outw[d] = sh ? (u32)((((u64)hi << 32) | lo) >> (8u * sh)) : lo;
        }
        const u32 done = headn + nwords * 4u;
        if (threadIdx.x < total - done) out[done + threadIdx.x] = stage[done + threadIdx.x];
    }
    if (EDGE && threadIdx.x == 0 && gt == (e - 1) / K1_TILE) nOut[b] = ebase + total; // the block's last tile
    if (threadIdx.x < 8) {
        u32 v = 0;
#pragma unroll
        for (int r = 0; r < 8; r++) v |= pres[threadIdx.x * 8 + r];
        if (v) atomicOr(&present[b * 8 + threadIdx.x], v);
    }
}

__global__ __launch_bounds__(256) void k1_emit(const u8 *__restrict__ in,
                                               const u32 *__restrict__ agg_head,
                                               const u32 *__restrict__ agg_c,
                                               const u32 *__restrict__ starts,
                                               const u32 *__restrict__ binfo, u8 *__restrict__ T,
                                               u32 *__restrict__ present, u32 *__restrict__ nOut, u32 S) {
    const u32 b = blockIdx.y;
    const u32 s = starts[b], e = starts[b + 1];
    const u32 gt = s / K1_TILE + blockIdx.x;
    if (gt * K1_TILE >= e) return;
    const u32 ebase = emit_base(in, agg_head, agg_c, binfo, b, s, gt);
    // inner tile: begins after s and ends before the block's last byte
    if (gt * K1_TILE > s && (gt + 1u) * K1_TILE < e) emit_tile<false>(in, agg_head, s, e, gt, ebase, b, T, present, nOut, S);
    else emit_tile<true>(in, agg_head, s, e, gt, ebase, b, T, present, nOut, S);
}

// ---------------------------------------------------------------- CRC
__constant__ u32 c_crc_table4[4][256]; // [k][v]: the register after byte v and k zero bytes
__constant__ u32 c_xpow16[257]; // x^(128*m) mod P, m = 0..256
__constant__ u32 c_xpow1[16];   // x^(8*r) mod P, r = 0..15

__device__ __forceinline__ u32 gfmul(u32 a, u32 b) {
    u32 r = 0;
#pragma unroll 8
    for (int i = 31; i >= 0; i--) {
        r = (r << 1) ^ ((r & 0x80000000u) ? 0x04c11db7u : 0u);
        if ((b >> i) & 1u) r ^= a;
    }
    return r;
}

// raw CRC (register from 0) of each 4096-byte tile of the block's input.  One wave per tile, 64
// contiguous bytes per lane: four table look-ups per 4 bytes (the four tables advance a byte by 0..3
// further zero bytes), then ONE multiplication by x^(8 * bytes behind the lane's piece) and an xor
// over the wave.  (16 bytes per lane and two 32-step multiplications each cost 26 instructions per
// byte, and this kernel runs beside k1_emit, which needs the same issue slots.)
#define K1_CRC_TPW 4 // tiles per workgroup
__global__ __launch_bounds__(256) void k1_crc_p1(const u8 *__restrict__ in,
                                                 const u32 *__restrict__ starts,
                                                 u32 *__restrict__ cagg, u32 ETPB) {
    const u32 b = blockIdx.y;
    const u32 s = starts[b], e = starts[b + 1];
    __shared__ u32 tab[4][256];
#pragma unroll
    for (int k = 0; k < 4; k++) tab[k][threadIdx.x] = c_crc_table4[k][threadIdx.x];
    __syncthreads();
    const u32 tile = blockIdx.x * K1_CRC_TPW + (u32)wave_id();
    const u32 base = s + tile * K1_TILE;
    if (tile * K1_TILE >= e - s) return;
    const u32 tlen = min((u32)K1_TILE, e - base);
    const u32 p0 = (u32)lane_id() * 64u;
    u32 reg = 0;
    if (p0 < tlen) {
        const u32 cnt = min(64u, tlen - p0);
        const u8 *a = in + base + p0;
        if (cnt == 64u) {
            // aligned words around the piece (the bytes outside it share a word with bytes inside)
            const u32 m = (u32)((uintptr_t)a & 3u); // the same in every lane
            const u32 *aw = (const u32 *)(a - m);
            u32 D[17];
#pragma unroll
            for (int k = 0; k < 16; k++) D[k] = aw[k];
            D[16] = m ? aw[16] : 0u;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const u32 wle = m == 0 ? D[k] : (m == 1 ? __builtin_amdgcn_alignbyte(D[k + 1], D[k], 1)
                                                 : (m == 2 ? __builtin_amdgcn_alignbyte(D[k + 1], D[k], 2)
                                                           : __builtin_amdgcn_alignbyte(D[k + 1], D[k], 3)));
                const u32 x = reg ^ __builtin_bswap32(wle);
                reg = tab[3][x >> 24] ^ tab[2][(x >> 16) & 255u] ^ tab[1][(x >> 8) & 255u] ^ tab[0][x & 255u];
            }
        } else {
            for (u32 q = 0; q < cnt; q++) reg = (reg << 8) ^ tab[0][((reg >> 24) ^ a[q]) & 0xff];
        }
        const u32 after = tlen - (p0 + cnt); // bytes after this piece inside the tile
        u32 mult = c_xpow16[after >> 4];
        if (after & 15u) mult = gfmul(mult, c_xpow1[after & 15u]);
        reg = gfmul(reg, mult);
    }
    // xor-reduce
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) reg ^= __shfl_xor(reg, d, 64);
    if (lane_id() == 0) cagg[(size_t)b * ETPB + tile] = reg;
}

// Combines the tile CRCs of a block: crc = (init * x^(8 len) + sum_t raw_t * x^(8 * bytes after tile t)) ^ ~0.
// One wave per block: every lane folds a contiguous stretch of tiles (Horner over whole tiles:
// one multiplication by x^(8*4096) per tile), then the 64 partial values are combined with the
// power that belongs to the bytes after each lane's stretch.
__device__ __forceinline__ u32 xpow_bytes(u32 nbytes) { // x^(8 nbytes) mod P, nbytes < 2^32
    // x^(8 nbytes) = (x^(8*4096))^(nbytes / 4096) * x^(8 (nbytes % 4096))
    u32 r = gfmul(c_xpow16[(nbytes & 4095u) >> 4], c_xpow1[nbytes & 15u]);
    u32 q = nbytes >> 12;
    u32 base = c_xpow16[256];
    while (q) {
        if (q & 1u) r = gfmul(r, base);
        base = gfmul(base, base);
        q >>= 1;
    }
    return r;
}

__global__ __launch_bounds__(64) void k1_crc_p2(const u32 *__restrict__ starts, const u32 *__restrict__ cagg,
                                                u32 *__restrict__ crc, u32 nb, u32 ETPB) {
    const u32 b = blockIdx.x;
    if (b >= nb) return;
    const int lane = lane_id();
    const u32 len = starts[b + 1] - starts[b];
    const u32 tiles = (len + K1_TILE - 1) / K1_TILE;
    const u32 per = (tiles + 63) / 64;
    const u32 t0 = min(tiles, (u32)lane * per), t1 = min(tiles, t0 + per);
    const u32 xfull = c_xpow16[256]; // x^(8*4096)
    // this lane's stretch as if it stood alone: every tile but the block's last is full
    u32 acc = 0;
    u32 bytes = 0; // bytes in [t0, t1)
    for (u32 t = t0; t < t1; t++) {
        const u32 tlen = min((u32)K1_TILE, len - t * K1_TILE);
        const u32 mult = (tlen == K1_TILE) ? xfull : gfmul(c_xpow16[tlen >> 4], c_xpow1[tlen & 15]);
        acc = gfmul(acc, mult) ^ cagg[(size_t)b * ETPB + t];
        bytes += tlen;
    }
    // shift by the bytes that follow the stretch
    const u32 before = min(len, t0 * K1_TILE);
    const u32 after = len - before - bytes;
    if (t1 > t0) acc = gfmul(acc, xpow_bytes(after));
    else acc = 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc ^= __shfl_xor(acc, d, 64);
    if (lane == 0) crc[b] = ~(acc ^ gfmul(0xffffffffu, xpow_bytes(len)));
}

// ---------------------------------------------------------------- drivers
static bool g_crc_tables_ready[64] = {false};

static u32 h_gfmul(u32 a, u32 b) {
    u32 r = 0;
    for (int i = 31; i >= 0; i--) {
        r = (r << 1) ^ ((r & 0x80000000u) ? 0x04c11db7u : 0u);
        if ((b >> i) & 1u) r ^= a;
    }
    return r;
}

void k1_init_tables(bz2hip_ctx *c) {
    if (c->device >= 0 && c->device < 64 && g_crc_tables_ready[c->device]) return;
    u32 tab[256], xp16[257], xp1[16];
    for (u32 i = 0; i < 256; i++) {
        u32 v = i << 24;
        for (int k = 0; k < 8; k++) v = (v & 0x80000000u) ? (v << 1) ^ 0x04c11db7u : (v << 1);
        tab[i] = v;
    }
    // x^8 mod P: polynomial "1" is register value 1; multiply by x eight times
    u32 x8 = 1;
    for (int k = 0; k < 8; k++) x8 = (x8 << 1) ^ ((x8 & 0x80000000u) ? 0x04c11db7u : 0u);
    xp1[0] = 1;
    for (int r = 1; r < 16; r++) xp1[r] = h_gfmul(xp1[r - 1], x8);
    u32 x128 = h_gfmul(xp1[15], x8);
    xp16[0] = 1;
    for (int m = 1; m <= 256; m++) xp16[m] = h_gfmul(xp16[m - 1], x128);
    {
        static u32 tab4[4][256];
        for (u32 i = 0; i < 256; i++) {
            tab4[0][i] = tab[i];
            for (int k = 1; k < 4; k++) tab4[k][i] = (tab4[k - 1][i] << 8) ^ tab[tab4[k - 1][i] >> 24];
        }
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_crc_table4), tab4, sizeof(tab4)));
    }
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_xpow16), xp16, sizeof(xp16)));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_xpow1), xp1, sizeof(xp1)));
    if (c->device >= 0 && c->device < 64) g_crc_tables_ready[c->device] = true;
}

// Scan a window of input: fills the cell records Hc, Pc (one per 256 bytes, up to the cell of
// position n), the tile-local 16-bit prefix P16[0..n] (c->w_P16) and the tile aggregates
// (asynchronous; independent of where blocks start).
// When the window is still on its way from the host (c->up: bz2hip_compress), it is uploaded in
// chunks on the copy stream and every chunk is scanned as soon as it has landed, so the transfer
// of chunk k + 1 runs under the scan of chunk k.
void run_scan(bz2hip_ctx *c, const u8 *d_in, u32 n, u32 *d_rs, u32 *d_P, u32 *d_agg1, u32 *d_agg2) {
    Batch &B = c->batch;
    const u32 tiles = (n + K1_TILE - 1) / K1_TILE;
    ProfScope ps(c, P_RLE1_SCAN, 0);
    const bool chunked = c->up.active && c->up.dev == d_in && c->up.n == n;
    const size_t chunk = chunked ? c->up.chunk : (size_t)n;
    u32 *carry_max = B.ctl + 10; // running last-head carry between chunks
    size_t k = 0;
    for (size_t off = 0; off < n; off += chunk, k++) {
        const size_t len = std::min<size_t>(chunk, n - off);
        const u32 t0 = (u32)(off / K1_TILE), t1 = (u32)((off + len + K1_TILE - 1) / K1_TILE);
        if (chunked) {
            HIPCHK(hipMemcpyAsync(c->up.dev + off, c->up.host + off, len, hipMemcpyHostToDevice, c->stream_copy));
            hipEvent_t e = c->up_event(k);
            HIPCHK(hipEventRecord(e, c->stream_copy));
            HIPCHK(hipStreamWaitEvent(c->stream, e, 0));
        }
        // n is passed as the bytes present so far: the last tile of a chunk clamps at its end, and
        // chunks are multiples of the tile, so every tile sees exactly its own bytes
        const u32 upto = (u32)(off + len);
        k1_heads_p1<<<t1 - t0, 256, 0, c->stream>>>(d_in, upto, c->w_agg1r, t0);
        KCHECK();
        k1_scan_tiles<true><<<(t1 - t0 + 1023) / 1024, 1024, 0, c->stream>>>(c->w_agg1r, d_agg1, t0, t1, carry_max);
        KCHECK();
        k1_cells<<<t1 - t0, 256, 0, c->stream>>>(d_in, upto, d_agg1, d_rs, c->w_agg2r, d_P, c->w_P16, t0);
        KCHECK();
    }
    c->up.active = false;
    k1_scan_tiles<false><<<std::max(1u, (tiles + 1023) / 1024), 1024, 0, c->stream>>>(c->w_agg2r, d_agg2, 0, tiles, B.ctl + 8);
    KCHECK();
    k1_sentinels<<<1, 64, 0, c->stream>>>(d_rs, d_P, c->w_P16, d_agg2, carry_max, n, B.ctl + 8);
    KCHECK();
}

// The serial block chain over a scanned window, from block start s0, for blocks that start below
// `stop`.  Returns the number of blocks found (<= max_blocks); h_starts receives count+1 offsets
// (h_starts[count] = end of the last block = start of the next).
u32 run_chain(bz2hip_ctx *c, const u8 *d_in, u32 n, int level, const u32 *d_rs, const u32 *d_P,
              const u32 *d_agg2, u32 *d_starts, u32 max_blocks, u32 s0, u32 stop,
              std::vector<u32> &h_starts) {
    Batch &B = c->batch;
    {
        ProfScope ps(c, P_RLE1_SPLIT, 0);
        // with more than a few blocks ahead, the brackets of all possible block ends first (parallel)
        const u32 bs = (u32)level * 100000u;
        const bool hints = c->w_hint && c->split_hints &&
                           (c->split_hints > 1 || (u64)(std::min(stop, n) - std::min(s0, n)) > 8ull * bs);
        if (hints) {
            k1_hints<<<(n / K1_CELL + 1 + 255) / 256, 256, 0, c->stream>>>(d_P, c->w_P16, d_agg2, n, bs, c->w_hint);
            KCHECK();
        }
        k1_split<<<1, 64, 0, c->stream>>>(d_in, d_rs, d_P, c->w_P16, d_agg2, n, bs, max_blocks, d_starts, c->w_binfo,
                                          B.ctl + 9, s0, stop, hints ? c->w_hint : (const u32 *)nullptr);
        KCHECK();
    }
    u32 cnt = 0;
    if ((size_t)max_blocks + 2 <= H_PIN_WORDS) {
        // one round trip: the count and every possible start go to page-locked scratch together
        // (a few KB; the second, dependent read-back cost more than the bytes)
        u32 *hp = c->h_pin;
        HIPCHK(hipMemcpyAsync(hp, B.ctl + 9, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(hp + 1, d_starts, sizeof(u32) * ((size_t)max_blocks + 1), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        cnt = hp[0];
        h_starts.assign(hp + 1, hp + 1 + cnt + 1);
        return cnt;
    }
    HIPCHK(hipMemcpyAsync(&cnt, B.ctl + 9, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    h_starts.resize(cnt + 1);
    HIPCHK(hipMemcpyAsync(h_starts.data(), d_starts, sizeof(u32) * (cnt + 1), hipMemcpyDeviceToHost,
                          c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return cnt;
}

u32 run_split(bz2hip_ctx *c, const u8 *d_in, u32 n, int level, u32 *d_rs, u32 *d_P, u32 *d_agg1,
              u32 *d_agg2, u32 *d_starts, u32 max_blocks, std::vector<u32> &h_starts) {
    run_scan(c, d_in, n, d_rs, d_P, d_agg1, d_agg2);
    return run_chain(c, d_in, n, level, d_rs, d_P, d_agg2, d_starts, max_blocks, 0, n, h_starts);
}

// CRC of nb byte ranges [starts[b], starts[b+1]) of d_data (the decoder's use of the CRC kernels)
void run_crc_blocks(bz2hip_ctx *c, const u8 *d_data, const u32 *d_starts, int nb, u32 max_len,
                    u32 *d_cagg, u32 ETPB, u32 *d_crc) {
    const u32 tiles = (max_len + K1_TILE - 1) / K1_TILE;
    if (tiles) {
        k1_crc_p1<<<dim3((tiles + K1_CRC_TPW - 1) / K1_CRC_TPW, nb), 256, 0, c->stream>>>(d_data, d_starts, d_cagg, ETPB);
        KCHECK();
    }
    k1_crc_p2<<<nb, 64, 0, c->stream>>>(d_starts, d_cagg, d_crc, nb, ETPB);
    KCHECK();
}

// RLE1-encode blocks [b0, b0+nb) of the window into the batch; CRC per block.
void run_rle1_emit(bz2hip_ctx *c, const u8 *d_in, const u32 *d_agg_head, const u32 *d_starts, u32 b0,
                   int nb, u32 max_in_len, u32 ETPB) {
    Batch &B = c->batch;
    const u32 tiles = (max_in_len + K1_TILE - 1) / K1_TILE;
    // the emit tiles are the scan's global tiles: a block of `len` bytes touches up to one more
    dim3 egrid(std::min(tiles + 1, ETPB), nb);
    // The block CRCs depend only on the input and the block starts: they run on the other stream,
    // next to the emit scan; this stream waits for them before anything that follows.
    {
        k1_crc_p1<<<dim3((tiles + K1_CRC_TPW - 1) / K1_CRC_TPW, nb), 256, 0, c->stream_other>>>(d_in, d_starts + b0, c->w_cagg, ETPB);
        KCHECK();
        k1_crc_p2<<<nb, 64, 0, c->stream_other>>>(d_starts + b0, c->w_cagg, B.crc, nb, ETPB);
        KCHECK();
        HIPCHK(hipEventRecord(c->crc_done, c->stream_other));
    }
    {
        ProfScope ps(c, P_RLE1_EMIT, 0);
        HIPCHK(hipMemsetAsync(B.present, 0, (size_t)nb * 8 * sizeof(u32), c->stream));
        k1_emit<<<egrid, 256, 0, c->stream>>>(d_in, d_agg_head, c->w_agg2, d_starts + b0,
                                              c->w_binfo + 2 * (size_t)b0, B.T, B.present, B.n, B.S);
        KCHECK();
    }
    HIPCHK(hipStreamWaitEvent(c->stream, c->crc_done, 0));
}
