// k3_mtf.hip -- Move-To-Front + RLE2 (RUNA/RUNB) stage, batched over blocks.
//
// Replaces BZip2MTFAndRLE2StageEncoder.encode() (BZip2MTFAndRLE2StageEncoder.java:50-134,
// S/BZip2MTFAndRLE2StageEncoder.swift:43-127) and MoveToFront.valueToFront
// (S/BZip2MoveToFront.swift:66-82).
//
// MTF is a sequential recurrence over the BWT string.  It is cut into chunks of MTF_CHUNK
// symbols, one wavefront per chunk, the 256-entry list held 4 entries per lane:
//   mtf_recency : last occurrence of every symbol in the chunk -> recency list (distinct symbols
//                 by last occurrence, latest first) and its length d
//   mtf_compose : one wave per block walks the chunks: entry_state(c+1) =
//                 recency(c) ++ (entry_state(c) minus recency(c))
//   mtf_replay  : re-run every chunk from its entry state (as last-use times) -> MTF position
//                 per BWT byte
// RLE2: zero runs (joined across tiles) become bijective base-2 RUNA/RUNB digits, positions
// p > 0 become symbol p+1, EOB is appended; output slots by a tile scan.
#include "ctx.hpp"

#define MTF_WG_CHUNKS 4

__device__ __forceinline__ u32 identity_list() {
    u32 l = lane_id() * 4;
    return l | ((l + 1) << 8) | ((l + 2) << 16) | ((l + 3) << 24);
}

// byte -> dense symbol map from the present bitmap (java:60-65); smap[256] in LDS
__device__ __forceinline__ void build_symbol_map(const u32 *__restrict__ presentWords, u8 *smap) {
    if (threadIdx.x < 256) {
        u32 v = threadIdx.x;
        u32 cnt = 0;
        for (u32 wi = 0; wi < (v >> 5); wi++) cnt += __popc(presentWords[wi]);
        cnt += __popc(presentWords[v >> 5] & ((1u << (v & 31)) - 1u));
        smap[v] = (u8)cnt;
    }
}

// Replay by timestamps instead of by list: every symbol keeps the time it was last moved to the
// front (4 symbols per lane: symbol v lives in lane v / 4, register v % 4); the MTF position of
// s is the number of symbols with a later time, i.e. 4 compares + 4 scalar popcounts, and the
// update is one register of one lane.  No list shifting: ~10 instructions per run head instead
// of ~25.  The entry state (a list) becomes times 255 - position.
// NSYM = dense symbol ids per lane (64 * NSYM >= alphabet size): 1, 2 or 4.  Symbols that do not
// occur sit at the end of every entry list (the recency operator only moves present symbols to the
// front) and keep the lowest times, so they never count towards a position: a block of <= 128
// symbols (text) needs 2 compares + 2 popcounts per run head, one of <= 64 (DNA, Markov) 1.

template <int NSYM>
__device__ __forceinline__ void mtf_replay_body(const u8 *__restrict__ src, u32 cbase, u32 cend,
                                                const u8 *smap, u32 *tsrow, u32 st,
                                                u8 *__restrict__ dst) {
    const int lane = lane_id();
    u8 rowv[MTF_CHUNK / 64];
#pragma unroll
    for (int r = 0; r < MTF_CHUNK / 64; r++) rowv[r] = src[cbase + r * 64 + lane];
    // entry list -> times
#pragma unroll
    for (int k = 0; k < 4; k++) tsrow[(st >> (8 * k)) & 255u] = 255u - (u32)(lane * 4 + k);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    u32 L[NSYM];
#pragma unroll
    for (int k = 0; k < NSYM; k++) L[k] = tsrow[lane * NSYM + k];
    // tsrow[] stays a mirror of the registers: the time of the head's symbol is read from it (one
    // uniform LDS read, issued one head ahead: consecutive heads are different symbols), so no
    // register has to be selected by the symbol; the update compares all ids of the lane with s
    const u32 id0 = (u32)lane * (u32)NSYM;
    typedef __attribute__((address_space(3))) u32 lds_u32;
    lds_u32 *const tsw = (lds_u32 *)tsrow;
    u32 prev = 0xffffffffu;
    u32 now = 256;
#pragma unroll
    for (int r = 0; r < MTF_CHUNK / 64; r++) {
        const u32 rowbase = cbase + r * 64;
        if (rowbase >= cend) break;
        const u32 j = rowbase + lane;
        const u32 sym = (j < cend) ? smap[rowv[r]] : 0;
        u32 mypos = 0;
        const u32 left = (u32)__builtin_amdgcn_update_dpp((int)prev, (int)sym, 0x138, 0xf, 0xf, false);
        u64 heads = __ballot(j < cend && sym != left);
        if (heads) {
            // one head: (iA, pA = &ts[sA], tA) is the head being placed; the next head's index,
            // mirror slot and time are fetched first.  Two copies with the roles swapped, so that
            // nothing has to be moved between iterations.
#define MTF_HEAD(iA, sA, pA, tA, iB, sB, pB, tB)                                                       \
    iA = __builtin_ctzll((unsigned long long)heads);                                                   \
    asm("s_bitset0_b64 %0, %1" : "+s"(heads) : "s"(iA));                                               \
    iB = __builtin_ctzll((unsigned long long)heads) & 63;                                              \
    sB = __builtin_amdgcn_readlane(sym, iB); /* lane 63 when no head is left: harmless */              \
    pB = tsw + sB;                                                                                     \
    tB = *pB;                                                                                          \
    {                                                                                                  \
        u32 pos = 0;                                                                                   \
        _Pragma("unroll") for (int k = 0; k < NSYM; k++) pos += (u32)__popcll(__ballot(L[k] > tA));    \
        _Pragma("unroll") for (int k = 0; k < NSYM; k++) L[k] = (id0 + (u32)k == sA) ? now : L[k];     \
        *pA = now; /* every lane, same slot, same value: cheaper than narrowing exec */               \
        mypos = write_lane(mypos, pos, iA); /* 1 VALU instead of move + compare + select */           \
        now++;                                                                                         \
    }
            int iA = __builtin_ctzll((unsigned long long)heads), iB;
            u32 sA = __builtin_amdgcn_readlane(sym, iA), sB;
            lds_u32 *pA = tsw + sA, *pB;
            u32 tA = *pA, tB;
            while (true) {
                MTF_HEAD(iA, sA, pA, tA, iB, sB, pB, tB)
                if (!heads) break;
                MTF_HEAD(iB, sB, pB, tB, iA, sA, pA, tA)
                if (!heads) break;
            }
#undef MTF_HEAD
        }
        prev = __builtin_amdgcn_readlane(sym, 63);
        if (j < cend) dst[j] = (u8)mypos;
    }
}

__global__ __launch_bounds__(256) void k3_mtf_replay(const u8 *__restrict__ bwt,
                                                     const u32 *__restrict__ nArr,
                                                     const u32 *__restrict__ present,
                                                     const u8 *__restrict__ mtfrec,
                                                     u8 *__restrict__ mtfpos, u32 S, u32 CPB) {
    const u32 b = blockIdx.y;
    const u32 n = nArr[b];
    __shared__ u8 smap[256];
    __shared__ u32 ts[MTF_WG_CHUNKS][256];
    build_symbol_map(present + b * 8, smap);
    __syncthreads();
    const int w = wave_id(), lane = lane_id();
    const u32 c = blockIdx.x * MTF_WG_CHUNKS + w;
    const u32 cbase = c * MTF_CHUNK;
    if (cbase >= n) return;
    const u8 *src = bwt + (size_t)b * S;
    const u32 cend = min(cbase + MTF_CHUNK, n);
    u32 uniq = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) uniq += __popc(present[b * 8 + q]);
    const u32 st = ((const u32 *)(mtfrec + ((size_t)b * CPB + c) * 256))[lane];
    u8 *dst = mtfpos + (size_t)b * S;
    if (uniq <= 64) mtf_replay_body<1>(src, cbase, cend, smap, &ts[w][0], st, dst);
    else if (uniq <= 128) mtf_replay_body<2>(src, cbase, cend, smap, &ts[w][0], st, dst);
    else mtf_replay_body<4>(src, cbase, cend, smap, &ts[w][0], st, dst);
}

// ---- Replay without a list and without a loop over run heads: MTF positions as inversion counts.
// Give every list entry a time: position t of the chunk has time t, the entry list's place i has time
// -1 - i.  For position t let prev(t) be the time of the previous occurrence of its symbol.  The
// symbols in front of it in the list are those whose last occurrence lies in (prev(t), t); every
// time in that interval belongs to exactly one occurrence, and an occurrence is NOT the last one of
// its symbol iff the symbol comes again at some c < t, i.e. iff prev(c) > prev(t) for a c < t:
//     position(t) = t - prev(t) - 1 - #{c < t : prev(c) > prev(t)}
// (a byte inside a run has prev = t - 1 and gets 0 by itself).  All prev(c) are distinct, so the count
// is "processed positions minus the rank of prev(t) among their prevs": a bitmap over the 1280
// possible times with per-word prefix counts, rebuilt after every 16 positions, plus the pairs
// inside those 16 by DPP row shifts.  A quarter wave (one DPP row of 16 lanes) owns a chunk, so a wave
// replays four chunks at once at ~100 instructions per 64 positions whatever the alphabet size; the
// loop over run heads above costs 17-33 per head.
#define MTI_CPW 4                       // chunks per wave (one per DPP row)
#define MTI_WG_CHUNKS (4 * MTI_CPW)     // chunks per workgroup of 4 waves

// (gfx9 has DPP operands on VOP1 / VOP2 only, not on compares: the condition comes out of a borrow)
#define MTI_ROW " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"

// number of earlier lanes of the 16-lane row whose value is greater (values < 2^32 - 1).  With a DPP
// operand v_sub_co computes "shifted lane minus own lane" (measured: the reversed opcode does the
// same), so the values are complemented: ~v[t-n] - ~v[t] borrows iff v[t-n] > v[t].  A lane without a
// source reads 0 and always borrows: 15 - l such steps for lane l of the row.
__device__ __forceinline__ u32 row16_greater_before(u32 v) {
    u32 acc = 0, x;
    const u32 w = ~v;
#define MTI_GT(n)                                                                                      \
    "v_sub_co_u32_dpp %1, vcc, %2, %2 row_shr:" #n MTI_ROW "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
    asm volatile("s_nop 1\n\t" MTI_GT(15) MTI_GT(14) MTI_GT(13) MTI_GT(12) MTI_GT(11) MTI_GT(10) MTI_GT(9) MTI_GT(8)
                 MTI_GT(7) MTI_GT(6) MTI_GT(5) MTI_GT(4) MTI_GT(3) MTI_GT(2) MTI_GT(1)
                 : "+v"(acc), "=&v"(x)
                 : "v"(w)
                 : "vcc");
#undef MTI_GT
    return acc - (15u - ((u32)lane_id() & 15u));
}

// inclusive sum over the 16-lane row
__device__ __forceinline__ u32 row16_incl_scan(u32 x) {
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);
    return x;
}

#define MTI_BUFSTRIDE (MTF_CHUNK + 16)  // LDS bytes per chunk: the four rows of a wave start 4 banks apart
#define MTI_LPSTRIDE 264                // table words per chunk (256 used): rows 8 banks apart

__global__ __launch_bounds__(256) void k3_mtf_replay_inv(const u8 *__restrict__ bwt,
                                                         const u32 *__restrict__ nArr,
                                                         const u32 *__restrict__ present,
                                                         const u8 *__restrict__ mtfrec,
                                                         u8 *__restrict__ mtfpos, u32 S, u32 CPB) {
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    const u32 b = blockIdx.y;
    const u32 n = nArr[b];
    __shared__ u8 smap[256];
    __shared__ __attribute__((aligned(16))) u8 buf[4][MTI_CPW * MTI_BUFSTRIDE]; // bytes in, positions out
    __shared__ u32 lastP[4][MTI_CPW][MTI_LPSTRIDE];            // byte -> time + 256 of its last occurrence (low half)
    __shared__ __attribute__((aligned(16))) u32 bmx[4][256];   // per chunk 64 bitmap words (40 used), lane l of the row owns 4 l .. 4 l + 3
    __shared__ __attribute__((aligned(16))) u32 bmy[4][256];   // bits set in the chunk's words before this one
    __shared__ u32 sink[4][64];                                // where lanes write that have nothing to write
    __shared__ u8 sinv[256];                                   // dense symbol -> byte (absent bytes take the ids behind the present ones)
    build_symbol_map(present + b * 8, smap);
    __syncthreads();
    {
        // The table below is indexed by the byte itself (no symbol lookup per position); only the entry
        // lists speak in dense symbols.
        const u32 v = threadIdx.x, uniq = (u32)smap[255] + ((present[b * 8 + 7] >> 31) & 1u);
        const bool here = (present[b * 8 + (v >> 5)] >> (v & 31u)) & 1u;
        sinv[here ? (u32)smap[v] : uniq + (v - (u32)smap[v])] = (u8)v;
    }
    __syncthreads();
    const int w = wave_id(), lane = lane_id();
    const u32 q = (u32)lane >> 4, l = (u32)lane & 15u;
    const u32 wbase = (blockIdx.x * MTI_WG_CHUNKS + (u32)w * MTI_CPW) * MTF_CHUNK; // the wave's 4 chunks are contiguous
    if (wbase >= n) return;
    const u8 *src = bwt + (size_t)b * S;
    u8 *dst = mtfpos + (size_t)b * S;
    const u32 c = wbase / MTF_CHUNK + q;
    const u32 cbase = c * MTF_CHUNK;
    const u32 cend = min(cbase + MTF_CHUNK, n); // < cbase: a chunk past the end of the block
    // the chunks' bytes: load r brings chunk r, 16 bytes per lane (S is a multiple of 4096: the loads
    // stay inside the block's slot)
#pragma unroll
    for (int r = 0; r < MTI_CPW; r++)
        *(u32x4 *)&buf[w][r * MTI_BUFSTRIDE + lane * 16] = *(const u32x4 *)(src + wbase + r * MTF_CHUNK + lane * 16);
    // entry list -> times of the symbols' last occurrences before the chunk (place i: 255 - i)
    if (cbase < n) {
        const u32x4 e = *(const u32x4 *)(mtfrec + ((size_t)b * CPB + c) * 256 + l * 16u);
#pragma unroll
        for (int k = 0; k < 16; k++) lastP[w][q][sinv[(e[k >> 2] >> (8 * (k & 3))) & 255u]] = 255u - (l * 16u + (u32)k);
    }
    ((u32x4 *)&bmx[w][0])[lane] = (u32x4){0u, 0u, 0u, 0u};
    ((u32x4 *)&bmy[w][0])[lane] = (u32x4){0u, 0u, 0u, 0u};
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    u32 *const lp = &lastP[w][q][0];
    u32 *const bx = &bmx[w][q * 64u], *const by = &bmy[w][q * 64u];
    u8 *const bq = &buf[w][q * MTI_BUFSTRIDE];
    u32 *const mysink = &sink[w][lane];
    // The LDS serves one wave's operations in order, so "insert, then read the words back" and "write
    // the prefix counts, then look one up" need no wait of their own (the compiler's waits for the
    // loaded values are enough); MTI_ORDER only keeps the compiler from moving them past each other.
    // What the LDS does charge for is lanes of one instruction on one bank: atomics of a run of equal
    // bytes all hit one word (the first version spent 77 % of its LDS cycles on conflicts), so runs
    // inside the 16 positions are folded before anything touches the LDS: only a run's last lane
    // marks the symbol, and a run's bits go into the bitmap as one range per word.
#define MTI_ORDER() asm volatile("" ::: "memory")
    const u32 mybit = 0x10000u << l, lowmask = (1u << l) - 1u;
    u32 symn = bq[l];
    for (u32 k = 0; k < MTF_CHUNK / 16; k++) {
        const u32 t = k * 16u + l;
        const bool valid = cbase + t < cend; // lanes past the end of the block change nothing
        const u32 sym = symn;
        // runs: a lane whose left neighbour in the row holds the same byte continues it (prev = t - 1)
        const u32 lsym = (u32)__builtin_amdgcn_update_dpp(0, (int)(sym + 1u), 0x111, 0xf, 0xf, true);
        const bool cont = valid & (lsym == sym + 1u);
        const u32 Cr = (u32)(__ballot(cont) >> (q * 16u)) & 0xffffu;  // the row's continuation lanes
        const bool tail = valid & (((Cr >> (l + 1u)) & 1u) == 0u);     // last lane of its run
        // Previous occurrence of the others.  The run tails mark themselves in the upper half of
        // their symbol's table word; a lane then sees the time from before this step (lower half) and
        // the tails of its symbol among the 16: the nearest one below it is its predecessor, and the
        // tail with none above it writes the new time (which also clears the marks).
        u32 *const slot = &lp[sym];
        atomicOr(tail ? slot : mysink, mybit);
        MTI_ORDER();
        const u32 v = *slot;
        const u32 nextb = bq[(t + 16u) & (MTF_CHUNK - 1u)]; // the next step's byte (the last step re-reads a position)
        const u32 m = v >> 16, lower = m & lowmask;
        const u32 pin = 256u + k * 16u + (31u - (u32)__clz(lower | 1u)); // the lane of the highest mark below this one
        const u32 Ph = lower ? pin : (v & 0xffffu);
        const u32 P = valid ? (cont ? 255u + t : Ph) : 0u;
        MTI_ORDER();
        *((tail & ((m >> (l + 1u)) == 0u)) ? slot : mysink) = 256u + t;
        symn = nextb;
        // processed positions with a greater prev: all 16 k of them minus those below P
        const u32 wi = P >> 5, pb = P & 31u;
        const u32 below = (u32)__popc(bx[wi] & ((1u << pb) - 1u)) + by[wi];
        const u32 inv = k * 16u - below + row16_greater_before(P);
        MTI_ORDER();
        bq[t] = (u8)(t + 255u - P - inv);
        // insert: a run head its own bit; of a run's continuation lanes the first one, and any whose bit
        // opens a word, the run's bits in that word
        const bool leader = cont & ((pb == 0u) | ((((Cr << 1) >> l) & 1u) == 0u)); // (no short circuit: no branch)
        const u32 rem = (u32)__builtin_ctz(~(Cr >> l));               // continuation lanes from this one on
        const u32 range = ((1u << min(rem, 32u - pb)) - 1u) << pb;     // (rem >= 1 for a continuation lane)
        const bool head = valid & !cont;
        atomicOr((head | leader) ? &bx[wi] : mysink, head ? 1u << pb : range);
        MTI_ORDER();
        // prefix counts again: lane l owns words 4 l .. 4 l + 3
        const u32x4 ww = ((const u32x4 *)&bmx[w][0])[lane];
        const u32 c0 = (u32)__popc(ww.x), c1 = c0 + (u32)__popc(ww.y), c2 = c1 + (u32)__popc(ww.z),
                  c3 = c2 + (u32)__popc(ww.w);
        const u32 ex = row16_incl_scan(c3) - c3;
        ((u32x4 *)&bmy[w][0])[lane] = (u32x4){ex, ex + c0, ex + c1, ex + c2};
        MTI_ORDER();
    }
#undef MTI_ORDER
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    // positions out (units that start inside the block)
#pragma unroll
    for (int r = 0; r < MTI_CPW; r++) {
        const u32 o = wbase + (u32)r * MTF_CHUNK + (u32)lane * 16u;
        if (o < n) *(u32x4 *)(dst + o) = *(const u32x4 *)&buf[w][r * MTI_BUFSTRIDE + lane * 16];
    }
}

// Recency list of a chunk without running the MTF: last occurrence of every symbol (LDS
// atomic max), then rank the present symbols by last occurrence, latest first.
__global__ __launch_bounds__(256) void k3_mtf_recency(const u8 *__restrict__ bwt,
                                                      const u32 *__restrict__ nArr,
                                                      const u32 *__restrict__ present,
                                                      u8 *__restrict__ mtfrec,
                                                      u32 *__restrict__ mtfcnt, u32 S, u32 CPB) {
    const u32 b = blockIdx.y;
    const u32 n = nArr[b];
    __shared__ u8 smap[256];
    __shared__ __attribute__((aligned(16))) u32 last[MTF_WG_CHUNKS][256];
    __shared__ __attribute__((aligned(16))) u8 recl[MTF_WG_CHUNKS][256];
    build_symbol_map(present + b * 8, smap);
    for (int q = threadIdx.x; q < MTF_WG_CHUNKS * 256; q += 256) (&last[0][0])[q] = 0;
    __syncthreads();
    const int w = wave_id(), lane = lane_id();
    const u32 c = blockIdx.x * MTF_WG_CHUNKS + w;
    const u32 cbase = c * MTF_CHUNK;
    if (cbase >= n) return;
    const u32 cend = min(cbase + MTF_CHUNK, n);
    const u8 *src = bwt + (size_t)b * S;
    // The recency list holds the chunk's symbols by last occurrence, latest first.  An element is
    // its symbol's last occurrence iff the LDS maximum equals its position; its place in the list
    // is the number of such elements after it: rows from the last to the first, ballot + popcount.
    u8 sy[MTF_CHUNK / 64];
    {
        u8 rowv[MTF_CHUNK / 64];
#pragma unroll
        for (int r = 0; r < MTF_CHUNK / 64; r++) rowv[r] = src[cbase + r * 64 + lane];
#pragma unroll
        for (int r = 0; r < MTF_CHUNK / 64; r++) {
            const u32 j = cbase + r * 64 + lane;
            sy[r] = smap[rowv[r]];
            // only the last lane of a run inside the row: the lanes before it would raise the same
            // word to smaller values (and lanes of one atomic on one word are served one by one)
            const u32 right = (u32)__builtin_amdgcn_update_dpp((int)0x100, (int)(u32)sy[r], 0x130, 0xf, 0xf, false);
            const bool covered = lane < 63 && right == (u32)sy[r] && j + 1 < cend;
            if (j < cend && !covered) atomicMax(&last[w][sy[r]], j - cbase + 1);
        }
    }
    // wave-local: LDS operations of one wave complete in order
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    u32 d = 0;
    const u64 after = lane == 63 ? 0ull : (~0ull << (lane + 1));
#pragma unroll
    for (int r = MTF_CHUNK / 64 - 1; r >= 0; r--) {
        const u32 j = cbase + r * 64 + lane;
        const bool fl = j < cend && last[w][sy[r]] == j - cbase + 1;
        const u64 m = __ballot(fl);
        if (fl) recl[w][d + (u32)__popcll(m & after)] = sy[r];
        d += (u32)__popcll(m);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    ((u32 *)(mtfrec + ((size_t)b * CPB + c) * 256))[lane] = ((const u32 *)&recl[w][0])[lane];
    if (lane == 0) mtfcnt[(size_t)b * CPB + c] = d;
}

// Recency lists -> entry states.  The operator "state after = recency list ++ (state before without
// those symbols)" is associative, so the serial walk over a block's ~880 chunks is cut in MTF_GROUPS
// groups: (A) every group folds its chunks into one recency list, (B) one wave per block walks the
// groups' lists, (C) every group walks its chunks from its entry state and stores each chunk's
// entry state in place.  Serial depth ~28 + 32 + 28 steps instead of ~880.
#define MTF_GROUPS 32

// st[0..len) := rec[0..d) ++ (st[0..len) without those symbols); returns the new length (one wave)
__device__ __forceinline__ u32 mtf_compose_step(u8 *st, u8 *nst, u8 *flag, u32 rec, u32 d, u32 len) {
    const int lane = lane_id();
    ((u32 *)flag)[lane] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        u32 p = lane * 4 + k;
        if (p < d) {
            u8 v = (u8)(rec >> (8 * k));
            flag[v] = 1;
            nst[p] = v;
        }
    }
    __syncthreads();
    u32 cur = ((u32 *)st)[lane];
    u32 keepcnt = 0;
    bool keep[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        keep[k] = ((u32)(lane * 4 + k) < len) && flag[(cur >> (8 * k)) & 255u] == 0;
        keepcnt += keep[k] ? 1u : 0u;
    }
    u32 inc = wave_incl_scan_add(keepcnt);
    u32 o = d + inc - keepcnt;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (keep[k]) nst[o++] = (u8)(cur >> (8 * k));
    }
    const u32 newlen = d + (u32)__builtin_amdgcn_readlane((int)inc, 63);
    __syncthreads();
    ((u32 *)st)[lane] = ((u32 *)nst)[lane];
    __syncthreads();
    return newlen;
}

// (A) one wave per (group, block): the group's chunks folded into one recency list
__global__ __launch_bounds__(64) void k3_mtf_group(const u32 *__restrict__ nArr,
                                                   const u8 *__restrict__ mtfrec,
                                                   const u32 *__restrict__ mtfcnt, u8 *__restrict__ grec,
                                                   u32 *__restrict__ glen, u32 CPB) {
    const u32 b = blockIdx.y, g = blockIdx.x;
    const u32 chunks = (nArr[b] + MTF_CHUNK - 1) / MTF_CHUNK;
    const u32 per = (chunks + MTF_GROUPS - 1) / MTF_GROUPS;
    const u32 c0 = g * per, c1 = min(chunks, c0 + per);
    const int lane = lane_id();
    __shared__ u8 st[256], nst[256], flag[256];
    ((u32 *)st)[lane] = 0;
    __syncthreads();
    u32 len = 0;
    constexpr int PF = 8;
    for (u32 cb = c0; cb < c1; cb += PF) {
        u32 recs[PF], ds[PF];
#pragma unroll
        for (int q = 0; q < PF; q++) {
            const u32 c = cb + q;
            recs[q] = 0;
            ds[q] = 0;
            if (c < c1) {
                recs[q] = ((const u32 *)(mtfrec + ((size_t)b * CPB + c) * 256))[lane];
                ds[q] = mtfcnt[(size_t)b * CPB + c];
            }
        }
#pragma unroll
        for (int q = 0; q < PF; q++) {
            if (cb + q >= c1) break;
            len = mtf_compose_step(st, nst, flag, recs[q], ds[q], len);
        }
    }
    ((u32 *)(grec + ((size_t)b * MTF_GROUPS + g) * 256))[lane] = ((u32 *)st)[lane];
    if (lane == 0) glen[b * MTF_GROUPS + g] = len;
}

// (B) one wave per block: entry state of every group (written over the group's list)
__global__ __launch_bounds__(64) void k3_mtf_gscan(u8 *__restrict__ grec, const u32 *__restrict__ glen) {
    const u32 b = blockIdx.x;
    const int lane = lane_id();
    __shared__ u8 st[256], nst[256], flag[256];
    ((u32 *)st)[lane] = identity_list();
    __syncthreads();
    u32 recs[MTF_GROUPS], ds[MTF_GROUPS];
#pragma unroll
    for (int g = 0; g < MTF_GROUPS; g++) {
        recs[g] = ((const u32 *)(grec + ((size_t)b * MTF_GROUPS + g) * 256))[lane];
        ds[g] = glen[b * MTF_GROUPS + g];
    }
#pragma unroll
    for (int g = 0; g < MTF_GROUPS; g++) {
        ((u32 *)(grec + ((size_t)b * MTF_GROUPS + g) * 256))[lane] = ((u32 *)st)[lane];
        (void)mtf_compose_step(st, nst, flag, recs[g], ds[g], 256);
    }
}

// (C) one wave per (group, block): turn the chunks' recency lists into entry states, in place
__global__ __launch_bounds__(64) void k3_mtf_compose(const u32 *__restrict__ nArr,
                                                     u8 *__restrict__ mtfrec,
                                                     const u32 *__restrict__ mtfcnt,
                                                     const u8 *__restrict__ grec, u32 CPB) {
    const u32 b = blockIdx.y, g = blockIdx.x;
    const u32 n = nArr[b];
    const u32 chunks = (n + MTF_CHUNK - 1) / MTF_CHUNK;
    const u32 per = (chunks + MTF_GROUPS - 1) / MTF_GROUPS;
    const u32 c0 = g * per, c1 = min(chunks, c0 + per);
    if (c0 >= c1) return;
    const int lane = lane_id();
    __shared__ u8 st[256];
    __shared__ u8 nst[256];
    __shared__ u8 flag[256];
    ((u32 *)st)[lane] = ((const u32 *)(grec + ((size_t)b * MTF_GROUPS + g) * 256))[lane];
    __syncthreads();
    // The walk is serial, but the recency lists do not depend on it: fetch them 8 chunks at a
    // time (independent loads) so that a step costs LDS time, not a global-memory round trip.
    constexpr int PF = 8;
    for (u32 cb = c0; cb < c1; cb += PF) {
        u32 recs[PF], ds[PF];
#pragma unroll
        for (int q = 0; q < PF; q++) {
            const u32 c = cb + q;
            recs[q] = 0;
            ds[q] = 0;
            if (c < c1) {
                recs[q] = ((const u32 *)(mtfrec + ((size_t)b * CPB + c) * 256))[lane];
                ds[q] = mtfcnt[(size_t)b * CPB + c];
            }
        }
#pragma unroll
        for (int q = 0; q < PF; q++) {
            const u32 c = cb + q;
            if (c >= c1) break;
            u32 *slot = (u32 *)(mtfrec + ((size_t)b * CPB + c) * 256);
            slot[lane] = ((u32 *)st)[lane]; // entry state of chunk c
            if (c + 1 == c1) break;
            (void)mtf_compose_step(st, nst, flag, recs[q], ds[q], 256);
        }
    }
}

// ------------------------------------------------------------------ RLE2
__device__ __forceinline__ u32 ndigits(u32 r) { return 31u - __clz(r + 1u); } // floor(log2(r+1))

// element j in [0, n]: j == n is the virtual EOB (non-zero)
// Both tile kernels give every thread 16 consecutive elements (one 16-byte load) and walk them in
// registers with "position after the last non-zero" as the only state; what enters a thread is a
// max-scan over the threads before it.  (A row of 64 elements per wave step cost 86 vector
// instructions per row in the emit kernel, which made it issue-bound: 0.45 ms for 0.5 GB.)
struct R2Lane {
    u32 w[4]; // the 16 positions
    u32 nz;   // bit k: element k is non-zero (the EOB counts, elements past it do not)
};
__device__ __forceinline__ void r2_load(const u8 *__restrict__ mp, u32 tbase, u32 n, R2Lane &L) {
    // the slot is padded to whole tiles and 16-byte aligned
    const uint4 v = *(const uint4 *)(mp + tbase);
    L.w[0] = v.x; L.w[1] = v.y; L.w[2] = v.z; L.w[3] = v.w;
    u32 nz = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        // a byte's bits folded onto its bit 0, the four bit-0s gathered by one multiplication
        u32 t = L.w[q] | (L.w[q] >> 4);
        t |= t >> 2;
        t |= t >> 1;
        nz |= (((t & 0x01010101u) * 0x01020408u) >> 24 & 15u) << (4 * q);
    }
    if (n < tbase + 16u) { // the EOB at n (non-zero) and nothing after it
        if (n >= tbase) nz = (nz & ((1u << (n - tbase)) - 1u)) | (1u << (n - tbase));
        else nz = 0;
    }
    L.nz = nz;
}
// position after the last non-zero before this thread inside the tile (0 = none), and the tile's
// per-wave last / first non-zero through wl / wf
__device__ __forceinline__ u32 r2_prev(const R2Lane &L, u32 tbase, u32 *wl, u32 *wf) {
    const int w = wave_id(), lane = lane_id();
    const u32 lastp1 = L.nz ? tbase + 32u - (u32)__clz(L.nz) : 0u;
    const u32 firstp = L.nz ? tbase + (u32)__ffs(L.nz) - 1u : 0xffffffffu;
    const u32 inc = wave_incl_scan_max(lastp1);
    const u32 fmin = ~wave_reduce_max(~firstp);
    if (lane == 63) wl[w] = inc;
    if (lane == 0) wf[w] = fmin;
    __syncthreads();
    u32 prevp1 = 0;
    for (int q = 0; q < w; q++) prevp1 = max(prevp1, wl[q]);
    const u32 up = __shfl_up(inc, 1, 64);
    if (lane > 0) prevp1 = max(prevp1, up);
    return prevp1;
}
// symbols the thread's elements emit when the last non-zero before them ended at prev (position + 1)
__device__ __forceinline__ u32 r2_count(const R2Lane &L, u32 tbase, u32 prev) {
    u32 cnt = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const bool nz = (L.nz >> k) & 1u;
        const u32 j = tbase + (u32)k;
        const u32 e = 1u + ndigits(j - prev);
        cnt += nz ? e : 0u;
        prev = nz ? j + 1u : prev;
    }
    return cnt;
}

// phase 1: tile summary {lz, tz, inner, allzero}; inner leaves out the run digits before the
// tile's first non-zero (that run may have begun in an earlier tile: phase 2 adds them)
__global__ __launch_bounds__(256) void k3_rle2_p1(const u8 *__restrict__ mtfpos,
                                                  const u32 *__restrict__ nArr,
                                                  u32 *__restrict__ r2agg, u32 S, u32 TPB) {
    const u32 b = blockIdx.y;
    const u32 n = nArr[b];
    const u32 base = blockIdx.x * R2TILE;
    if (base > n) return;
    const u32 end = min(base + R2TILE, n + 1);
    __shared__ u32 wl[4], wf[4], ws[4];
    const u32 tbase = base + threadIdx.x * 16u;
    R2Lane L;
    r2_load(mtfpos + (size_t)b * S, tbase, n, L);
    const u32 prevp1 = r2_prev(L, tbase, wl, wf);
    // the tile's first non-zero counts as if no zero preceded it
    const u32 cnt = r2_count(L, tbase, prevp1 ? prevp1 : tbase + (u32)__ffs(L.nz) - 1u);
    const u32 wsum = wave_reduce_add(cnt);
    if (lane_id() == 0) ws[wave_id()] = wsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        const u32 tot = ws[0] + ws[1] + ws[2] + ws[3];
        const u32 tl = max(max(wl[0], wl[1]), max(wl[2], wl[3]));
        const u32 tf = min(min(wf[0], wf[1]), min(wf[2], wf[3]));
        const u32 cnt_t = end - base;
        const u32 az = (tf == 0xffffffffu) ? 1u : 0u;
        u32 *o = r2agg + ((size_t)b * TPB + blockIdx.x) * 4;
        o[0] = az ? cnt_t : (tf - base);
        o[1] = az ? cnt_t : (end - tl);
        o[2] = tot;
        o[3] = az;
    }
}

// phase 2: one workgroup per block; serial walk over tile summaries -> {offset, carry_in}
__global__ __launch_bounds__(256) void k3_rle2_p2(const u32 *__restrict__ nArr,
                                                  const u32 *__restrict__ present,
                                                  u32 *__restrict__ r2agg, u32 *__restrict__ mtflen,
                                                  u32 *__restrict__ alpha, u32 *__restrict__ freq,
                                                  u32 TPB) {
    const u32 b = blockIdx.x;
    const u32 n = nArr[b];
    const u32 tiles = n / R2TILE + 1; // elements 0..n
    __shared__ u32 sm[256 * 4];
    u32 *g = r2agg + (size_t)b * TPB * 4;
    for (u32 i = threadIdx.x; i < 264; i += 256) freq[b * 264 + i] = 0;
    u32 off = 0, carry = 0;
    for (u32 t0 = 0; t0 < tiles; t0 += 256) {
        u32 cnt = min(256u, tiles - t0);
        __syncthreads();
        for (u32 i = threadIdx.x; i < cnt * 4; i += 256) sm[i] = g[(size_t)t0 * 4 + i];
        __syncthreads();
        if (threadIdx.x == 0) {
            for (u32 t = 0; t < cnt; t++) {
                u32 lz = sm[t * 4], tz = sm[t * 4 + 1], inner = sm[t * 4 + 2], az = sm[t * 4 + 3];
                sm[t * 4] = off;
                sm[t * 4 + 1] = carry;
                if (az) {
                    carry += lz;
                } else {
                    u32 r = carry + lz;
                    off += inner + (r > 0 ? ndigits(r) : 0u);
                    carry = tz;
                }
            }
        }
        __syncthreads();
        for (u32 i = threadIdx.x; i < cnt * 4; i += 256) g[(size_t)t0 * 4 + i] = sm[i];
    }
    if (threadIdx.x == 0) {
        mtflen[b] = off;
        u32 uniq = 0;
        for (int q = 0; q < 8; q++) uniq += __popc(present[b * 8 + q]);
        alpha[b] = uniq + 2;
    }
}

// phase 3: write symbols + frequencies.  The tile's symbols are put together in LDS (every thread
// at its own offset) and leave in coalesced 2-byte rows.
__global__ __launch_bounds__(256) void k3_rle2_p3(const u8 *__restrict__ mtfpos,
                                                  const u32 *__restrict__ nArr,
                                                  const u32 *__restrict__ r2agg,
                                                  const u32 *__restrict__ alpha,
                                                  u16 *__restrict__ mtf, u32 *__restrict__ freq,
                                                  u32 S, u32 S2, u32 TPB) {
    const u32 b = blockIdx.y;
    const u32 n = nArr[b];
    const u32 base = blockIdx.x * R2TILE;
    if (base > n) return;
    __shared__ u32 wl[4], wf[4], sm[8];
    // every counter four times (lane mod 4): RUNA, RUNB and the first positions are most of the symbols,
    // and lanes of one LDS atomic that meet on a word are served one by one
    __shared__ u32 hist[264 * 4];
    // symbols of the tile: at most one per element, plus the digits of a run that came in (< 20)
    __shared__ u16 sbuf[R2TILE + 32];
    for (u32 i = threadIdx.x; i < 264 * 4; i += 256) hist[i] = 0;
    u32 *const myhist = hist + (threadIdx.x & 3u);
    const u32 *ag = r2agg + ((size_t)b * TPB + blockIdx.x) * 4;
    const u32 tile_off = ag[0], carry_in = ag[1];
    const u32 eob = alpha[b] - 1;
    const u32 tbase = base + threadIdx.x * 16u;
    R2Lane L;
    r2_load(mtfpos + (size_t)b * S, tbase, n, L);
    const u32 prevp1 = r2_prev(L, tbase, wl, wf);
    // nothing non-zero before this thread in the tile: the run reaches the tile start and carry_in
    // zeros beyond it (prev may wrap below zero: only differences are used)
    u32 prev = prevp1 ? prevp1 : base - carry_in;
    const u32 cnt = r2_count(L, tbase, prev);
    u32 total;
    u32 o = block_excl_scan_add<256>(cnt, sm, &total);
    u32 runa = 0, runb = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if ((L.nz >> k) & 1u) {
            const u32 j = tbase + (u32)k;
            const u32 r = j - prev;
            prev = j + 1u;
            if (r > 0) { // java:83-99
                u32 rep = r - 1;
                for (;;) {
                    const u32 dgt = rep & 1u;
                    sbuf[o++] = (u16)dgt;
                    runb += dgt;
                    runa += dgt ^ 1u;
                    if (rep <= 1) break;
                    rep = (rep - 2) >> 1;
                }
            }
            const u32 v = (L.w[k >> 2] >> (8 * (k & 3))) & 255u;
            const u32 sym = (j == n) ? eob : (v + 1u);
            sbuf[o++] = (u16)sym;
            atomicAdd(&myhist[sym * 4u], 1u);
        }
    }
    if (runa) atomicAdd(&myhist[0], runa);
    if (runb) atomicAdd(&myhist[4], runb);
    __syncthreads();
    u16 *out = mtf + (size_t)b * S2 + tile_off;
    for (u32 q = threadIdx.x; q < total; q += 256) out[q] = sbuf[q];
    for (u32 i = threadIdx.x; i < 264; i += 256) {
        u32 c = hist[i * 4] + hist[i * 4 + 1] + hist[i * 4 + 2] + hist[i * 4 + 3];
        if (c) atomicAdd(&freq[b * 264 + i], c);
    }
}

void run_mtf(bz2hip_ctx *c, int nb, u32 maxn) {
    Batch &B = c->batch;
    if (nb == 0) return;
    const u32 CPB = B.S / MTF_CHUNK;
    const u32 chunks = (maxn + MTF_CHUNK - 1) / MTF_CHUNK;
    dim3 gridc((chunks + MTF_WG_CHUNKS - 1) / MTF_WG_CHUNKS, nb);
    {
        ProfScope ps(c, P_MTF_STATE);
        k3_mtf_recency<<<gridc, 256, 0, c->stream>>>(B.bwt, B.n, B.present, B.mtfrec, B.mtfcnt, B.S, CPB);
        KCHECK();
    }
    {
        ProfScope ps(c, P_MTF_COMPOSE);
        k3_mtf_group<<<dim3(MTF_GROUPS, nb), 64, 0, c->stream>>>(B.n, B.mtfrec, B.mtfcnt, B.mtfgrp, B.mtfglen, CPB);
        KCHECK();
        k3_mtf_gscan<<<nb, 64, 0, c->stream>>>(B.mtfgrp, B.mtfglen);
        KCHECK();
        k3_mtf_compose<<<dim3(MTF_GROUPS, nb), 64, 0, c->stream>>>(B.n, B.mtfrec, B.mtfcnt, B.mtfgrp, CPB);
        KCHECK();
    }
    {
        ProfScope ps(c, P_MTF_REPLAY, 2.0 * (double)maxn * nb); // read bwt 1 + write pos 1 (upper bound: slots)
        if (c->mtf_inv) {
            const dim3 gridi((chunks + MTI_WG_CHUNKS - 1) / MTI_WG_CHUNKS, nb);
            k3_mtf_replay_inv<<<gridi, 256, 0, c->stream>>>(B.bwt, B.n, B.present, B.mtfrec, B.mtfpos, B.S, CPB);
        } else {
            k3_mtf_replay<<<gridc, 256, 0, c->stream>>>(B.bwt, B.n, B.present, B.mtfrec, B.mtfpos, B.S, CPB);
        }
        KCHECK();
    }
    {
        ProfScope ps(c, P_RLE2);
        const u32 tiles = maxn / R2TILE + 1;
        dim3 grid(tiles, nb);
        k3_rle2_p1<<<grid, 256, 0, c->stream>>>(B.mtfpos, B.n, B.r2agg, B.S, B.TPB);
        KCHECK();
        k3_rle2_p2<<<nb, 256, 0, c->stream>>>(B.n, B.present, B.r2agg, B.mtflen, B.alpha, B.freq,
                                              B.TPB);
        KCHECK();
        k3_rle2_p3<<<grid, 256, 0, c->stream>>>(B.mtfpos, B.n, B.r2agg, B.alpha, B.mtf, B.freq, B.S,
                                                B.S + 64, B.TPB);
        KCHECK();
    }
}
