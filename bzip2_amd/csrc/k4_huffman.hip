// k4_huffman.hip -- Huffman stage: table count, seeds, 4 optimisation passes, code lengths,
// canonical codes, header / selector / table serialisation and symbol bit packing.
//
// Replaces BZip2HuffmanStageEncoder (S/BZip2HuffmanStageEncoder.swift:82-98,107-135,144-181,
// 192-248,254-287,294-329,336-353,360-373), HuffmanAllocator (S/BZip2HuffmanAllocator.swift:
// 44-223, Java-original loop headers), the block header and symbol map of
// BZip2BlockCompressor.close()/writeSymbolMap (S/BZip2BlockCompressor.swift:89-144,273-280) and
// the bit packing of BZip2BitOutputStream (S/BZip2BitOutputStream.swift:53-120).
#include "ctx.hpp"

#define HL 264 // padded alphabet stride

// ------------------------------------------------------------------ seeds (:82-98, :144-181)
__global__ void k4_seed(const u32 *__restrict__ mtflen, const u32 *__restrict__ alphaArr,
                        const u32 *__restrict__ freq, u32 *__restrict__ ntables,
                        u32 *__restrict__ nsel, u8 *__restrict__ hlen, u32 nb) {
    const u32 b = blockIdx.x;
    if (b >= nb) return;
    const u32 L = mtflen[b];
    const int alpha = (int)alphaArr[b];
    int tables = L >= 2400 ? 6 : L >= 1200 ? 5 : L >= 600 ? 4 : L >= 200 ? 3 : 2;
    __shared__ int lo[MAX_TABLES], hi[MAX_TABLES];
    if (threadIdx.x == 0) {
        ntables[b] = tables;
        nsel[b] = (L + GROUP - 1) / GROUP;
        const u32 *f = freq + b * HL;
        int remaining = (int)L, low_end = -1;
        for (int i = 0; i < tables; i++) {
            int target = remaining / (tables - i);
            int low_start = low_end + 1;
            int actual = 0;
            while (actual < target && low_end < alpha - 1) actual += (int)f[++low_end];
            // D23: Java-original order (subtract the dropped symbol's frequency, then step back); the
            // Swift text :164-167 steps back first and subtracts the new end's frequency.  The contract
            // follows the Java original.  PARITY UNPINNED at this statement:
            // nothing runnable decides between the two readings.  -DBZ2HIP_D23_SWIFT_TEXT flips it (the
            // expected stream under that reading is frozen in tests/golden as `d23_backoff_swift_text`).
            if (low_end > low_start && i != 0 && i != tables - 1 && ((tables - i) & 1) == 0) {
#ifdef BZ2HIP_D23_SWIFT_TEXT
                low_end--;
                actual -= (int)f[low_end];
#else
                actual -= (int)f[low_end--];
#endif
            }
            lo[i] = low_start;
            hi[i] = low_end;
            remaining -= actual;
        }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < MAX_TABLES * HL; q += blockDim.x) {
        int t = q / HL, j = q % HL;
        u8 v = 0;
        if (t < tables && j < alpha) v = (j < lo[t] || j > hi[t]) ? 15 : 0;
        hlen[(size_t)b * MAX_TABLES * HL + q] = v;
    }
}

// ------------------------------------------------------------------ selection pass (:192-242)
// grid (HS_PART_MAX, nb); each workgroup handles HS_GROUPS_PER_WG groups of 50 symbols
#define HS_SUB 256 // groups staged per sub-batch
__global__ __launch_bounds__(256) void k4_select(const u16 *__restrict__ mtf,
                                                 const u32 *__restrict__ mtflen,
                                                 const u32 *__restrict__ ntables,
                                                 const u8 *__restrict__ hlen,
                                                 u32 *__restrict__ hpart, u8 *__restrict__ sel,
                                                 u32 S2, int store) {
    const u32 b = blockIdx.y;
    const u32 L = mtflen[b];
    const u32 ngroups = (L + GROUP - 1) / GROUP;
    const u32 g0 = blockIdx.x * HS_GROUPS_PER_WG;
    if (g0 >= ngroups) return;
    const int tables = (int)ntables[b];
    __shared__ u64 plen[HL];
    // every counter four times (lane mod 4): the symbols are few and skewed (RUNA, RUNB, position 1),
    // and lanes of one LDS atomic that meet on a word are served one by one
    constexpr u32 HSR = 4;
    __shared__ u32 hist[MAX_TABLES * HL * HSR];
    __shared__ u32 stage[HS_SUB * GROUP / 2];
    for (u32 j = threadIdx.x; j < HL; j += 256) {
        u64 p = 0;
        for (int t = 0; t < MAX_TABLES; t++)
            p |= (u64)hlen[((size_t)b * MAX_TABLES + t) * HL + j] << (10 * t);
        plen[j] = p;
    }
    for (u32 q = threadIdx.x; q < MAX_TABLES * HL * HSR; q += 256) hist[q] = 0;
    const u32 hr = threadIdx.x & (HSR - 1u);
    const u16 *src = mtf + (size_t)b * S2;
    const u16 *st16 = (const u16 *)stage;
    for (u32 sub = 0; sub < HS_GROUPS_PER_WG / HS_SUB; sub++) {
        const u32 gs = g0 + sub * HS_SUB;
        if (gs >= ngroups) break;
        const u32 sbase = gs * GROUP; // even => 4-byte aligned
        const u32 scount = min((u32)(HS_SUB * GROUP), L - sbase);
        __syncthreads();
        const u32 *src32 = (const u32 *)(src + sbase);
        for (u32 q = threadIdx.x; q < (scount + 1) / 2; q += 256) stage[q] = src32[q];
        __syncthreads();
        const u32 g = gs + threadIdx.x;
        if (g < ngroups) {
            const u32 lbase = threadIdx.x * GROUP;
            const u32 cnt = min((u32)GROUP, L - g * GROUP);
            u64 cost = 0;
            u32 sw[GROUP / 2]; // a whole group: its 25 words and their 50 length words are read in two batches
            const bool whole = cnt == (u32)GROUP;
            if (whole) {
#pragma unroll
                for (int i = 0; i < GROUP / 2; i++) sw[i] = stage[threadIdx.x * (GROUP / 2) + i];
#pragma unroll
                for (int i = 0; i < GROUP / 2; i++) cost += plen[sw[i] & 0xffffu] + plen[sw[i] >> 16];
            } else {
                for (u32 i = 0; i < cnt; i++) cost += plen[st16[lbase + i]];
            }
            int best = 0;
            u32 bestc = (u32)(cost & 1023u);
            for (int t = 1; t < tables; t++) {
                u32 cc = (u32)((cost >> (10 * t)) & 1023u);
                if (cc < bestc) {
                    bestc = cc;
                    best = t;
                }
            }
            u32 *const hb = &hist[(u32)best * HL * HSR + hr];
            u32 prev = st16[lbase], run = 1; // cnt >= 1; equal neighbours are added at once
            if (whole) {
#pragma unroll
                for (int i = 1; i < GROUP; i++) {
                    const u32 sy = (i & 1) ? (sw[i >> 1] >> 16) : (sw[i >> 1] & 0xffffu);
                    if (sy != prev) {
                        atomicAdd(&hb[prev * HSR], run);
                        prev = sy;
                        run = 0;
                    }
                    run++;
                }
            } else {
                for (u32 i = 1; i < cnt; i++) {
                    const u32 sy = st16[lbase + i];
                    if (sy != prev) {
                        atomicAdd(&hb[prev * HSR], run);
                        prev = sy;
                        run = 0;
                    }
                    run++;
                }
            }
            atomicAdd(&hb[prev * HSR], run);
            if (store) sel[(size_t)b * SEL_STRIDE + g] = (u8)best;
        }
    }
    __syncthreads();
    u32 *out = hpart + ((size_t)b * HS_PART_MAX + blockIdx.x) * MAX_TABLES * HL;
    for (u32 q = threadIdx.x; q < MAX_TABLES * HL; q += 256)
        out[q] = hist[q * HSR] + hist[q * HSR + 1] + hist[q * HSR + 2] + hist[q * HSR + 3];
}

#ifdef EXP_TIMING
__device__ unsigned long long g_k4phase[8];
extern "C" void bz2hip_debug_k4(unsigned long long *out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_k4phase), sizeof(unsigned long long) * 8);
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_k4phase), z, sizeof(z));
}
#define K4_MARK(k) do { if (threadIdx.x == 0) { unsigned long long t_ = wall_clock64(); atomicAdd(&g_k4phase[k], t_ - k4_t); k4_t = t_; } } while (0)
#define K4_INIT unsigned long long k4_t = wall_clock64();
#else
#define K4_MARK(k)
#define K4_INIT
#endif
// ------------------------------------------------------------------ allocator (single lane)
// S/BZip2HuffmanAllocator.swift:44-66
// a[i] % len for a parent pointer: the pointers are `tail` or `tail + len` with tail < len, so one
// conditional subtraction is the remainder (an integer division costs ~40 instructions on a
// single lane's serial path); the division stays for any other value
__device__ __forceinline__ int ha_mod(int x, int len) {
    if ((unsigned)x < (unsigned)len) return x;
    if ((unsigned)x < 2u * (unsigned)len) return x - len;
    return x % len;
}

__device__ int ha_first(const int *a, int len, int i, int nodes_to_move) {
    int limit = i;
    int k = len - 2;
    while (i >= nodes_to_move && ha_mod(a[i], len) > limit) {
        k = i;
        i -= (limit - i + 1);
    }
    i = max(nodes_to_move - 1, i);
    while (k > i + 1) {
        int temp = (i + k) >> 1;
        if (ha_mod(a[temp], len) > limit) k = temp; else i = temp;
    }
    return k;
}

// S/BZip2HuffmanAllocator.swift:197-223 with :72-104, :112-122, :128-147, :155-189
#ifdef EXP_TIMING
#define K4A_MARK(k) do { unsigned long long t_ = wall_clock64(); atomicAdd(&g_k4phase[k], t_ - k4a_t); k4a_t = t_; } while (0)
#define K4A_INIT k4a_t = wall_clock64();
#else
#define K4A_MARK(k)
#define K4A_INIT
#endif
__device__ void ha_allocate(int *a, int len, int max_len) {
#ifdef EXP_TIMING
    unsigned long long k4a_t = 0;
#endif
    if (len == 2) { a[1] = 1; a[0] = 1; return; }
    if (len == 1) { a[0] = 1; return; }
    {   // setExtendedParentPointers
        K4A_INIT
        a[0] += a[1];
        int head = 0, tail = 1, top = 2;
        // One LDS round trip per merge: the two candidates of either queue are read together, the two
        // picks are selects on registers (as written in the reference, every pick is a dependent read
        // behind a branch: four round trips and eight branches per merge on one lane).  Reads past the
        // picks' guards (a[top], a[top + 1] with top >= len; a[head + 1] == a[tail]) land inside the
        // array and are not used.
        while (tail < len - 1) {
            const int hv0 = a[head], hv1 = a[head + 1];
            const int tv0 = a[top], tv1 = a[top + 1];
            const bool c1 = (top >= len) | (hv0 < tv0);
            int temp = c1 ? hv0 : tv0;
            if (c1) a[head] = tail;
            const int nh = c1 ? hv1 : hv0, nt = c1 ? tv0 : tv1;
            head += c1 ? 1 : 0;
            top += c1 ? 0 : 1;
            const bool c2 = (top >= len) | ((head < tail) & (nh < nt));
            temp += c2 ? nh : nt;
            if (c2) a[head] = tail + len;
            head += c2 ? 1 : 0;
            top += c2 ? 0 : 1;
            a[tail++] = temp;
        }
        K4A_MARK(4);
    }
    int r;
    {   // findNodesToRelocate
        int node = len - 2;
        for (int depth = 1; depth < max_len - 1 && node > 1; depth++) node = ha_first(a, len, node - 1, 0);
        r = node;
        K4A_MARK(5);
    }
    if (ha_mod(a[0], len) >= r) {
        int first_node = len - 2, next_node = len - 1;
        for (int depth = 1, avail = 2; avail > 0; depth++) {
            int last_node = first_node;
            first_node = ha_first(a, len, last_node - 1, 0);
            for (int i = avail - (last_node - first_node); i > 0; i--) a[next_node--] = depth;
            avail = (last_node - first_node) << 1;
        }
    } else {
        int insert_depth = max_len - (32 - __clz((unsigned)(r - 1)));
        int nodes_to_move = r;
        int first_node = len - 2, next_node = len - 1;
        int depth = (insert_depth == 1) ? 2 : 1;
        int left = (insert_depth == 1) ? nodes_to_move - 2 : nodes_to_move;
        for (int avail = depth << 1; avail > 0; depth++) {
            int last_node = first_node;
            first_node = (first_node <= nodes_to_move) ? first_node
                                                       : ha_first(a, len, last_node - 1, nodes_to_move);
            int offset = 0;
            if (depth >= insert_depth) {
                offset = min(left, 1 << (depth - insert_depth));
            } else if (depth == insert_depth - 1) {
                offset = 1;
                if (a[first_node] == last_node) first_node++;
            }
            for (int i = avail - (last_node - first_node + offset); i > 0; i--) a[next_node--] = depth;
            left -= offset;
            avail = (last_node - first_node + offset) << 1;
        }
    }
}

// ------------------------------------------------------------------ code lengths (:107-135, :245-247)
// grid (6, nb): sum the partial histograms, sort (freq<<9)|idx, allocate, scatter back
// NT threads per table: 256 when every table's workgroup is resident anyway (the histogram sum and the
// rank sort then take 256 lanes), 64 for many blocks (-1: 16 000 tables; three of four waves leave before
// the allocator, but until then they hold the slots other tables wait for)
template <int NT>
__global__ __launch_bounds__(NT) void k4_lengths(const u32 *__restrict__ mtflen,
                                                  const u32 *__restrict__ alphaArr,
                                                  const u32 *__restrict__ ntables,
                                                  const u32 *__restrict__ hpart,
                                                  u8 *__restrict__ hlen, u32 *__restrict__ hfreq) {
    const u32 b = blockIdx.y, t = blockIdx.x;
    if (t >= ntables[b]) return;
    const int alpha = (int)alphaArr[b];
    const u32 ngroups = (mtflen[b] + GROUP - 1) / GROUP;
    const u32 parts = (ngroups + HS_GROUPS_PER_WG - 1) / HS_GROUPS_PER_WG;
    __shared__ int key[HL];
    __shared__ int sorted[HL];
    __shared__ int a[HL];
    K4_INIT
    for (int j = threadIdx.x; j < HL; j += NT) {
        u32 f = 0;
        if (j < alpha) {
            // all parts requested at once (a loop to `parts` waits for every load in turn)
            u32 v[HS_PART_MAX];
#pragma unroll
            for (u32 p = 0; p < HS_PART_MAX; p++)
                v[p] = hpart[(((size_t)b * HS_PART_MAX + min(p, parts - 1u)) * MAX_TABLES + t) * HL + j];
#pragma unroll
            for (u32 p = 0; p < HS_PART_MAX; p++) f += p < parts ? v[p] : 0u;
        }
        hfreq[((size_t)b * MAX_TABLES + t) * HL + j] = f;
        key[j] = (int)((f << 9) | (u32)j);
    }
    __syncthreads();
    K4_MARK(0);
    for (int j = threadIdx.x; j < alpha; j += NT) {
        int kj = key[j];
        int rnk = 0;
        for (int i = 0; i < alpha; i++) rnk += (key[i] < kj) ? 1 : 0;
        sorted[rnk] = kj;
        a[rnk] = (int)((u32)kj >> 9);
    }
    __syncthreads();
    // The allocator is one lane's work (~6 k dependent instructions): the other three waves end here and
    // give their slots to the next tables (2600 level-1 blocks are 15 600 tables: the kernel is bound by
    // how many of these workgroups a CU holds, not by the allocator's latency).
    if (wave_id() != 0) return;
    K4_MARK(1);
    if (threadIdx.x == 0) ha_allocate(a, alpha, 20);
    K4_MARK(2);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    for (int i = threadIdx.x; i < alpha; i += 64)
        hlen[((size_t)b * MAX_TABLES + t) * HL + (sorted[i] & 0x1ff)] = (u8)a[i];
    K4_MARK(3);
}

// ------------------------------------------------------------------ canonical codes (:254-287)
// grid (6, nb).  Also per-table serialised size and data bits.
__global__ __launch_bounds__(256) void k4_codes(const u32 *__restrict__ alphaArr,
                                                const u32 *__restrict__ ntables,
                                                const u8 *__restrict__ hlen,
                                                const u32 *__restrict__ hfreq,
                                                u32 *__restrict__ hcode, u64 *__restrict__ tblbits) {
    const u32 b = blockIdx.y, t = blockIdx.x;
    u64 *tb = tblbits + ((size_t)b * MAX_TABLES + t) * 2;
    if (t >= ntables[b]) {
        if (threadIdx.x == 0) { tb[0] = 0; tb[1] = 0; }
        return;
    }
    const int alpha = (int)alphaArr[b];
    __shared__ int len[HL];
    __shared__ int cnt[32];
    __shared__ int firstc[32];
    __shared__ u32 red[8];
    if (threadIdx.x < 32) cnt[threadIdx.x] = 0;
    __syncthreads();
    const u8 *hl = hlen + ((size_t)b * MAX_TABLES + t) * HL;
    for (int j = threadIdx.x; j < alpha; j += 256) {
        int l = hl[j];
        len[j] = l;
        atomicAdd(&cnt[l & 31], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int mn = 32, mx = 0;
        for (int l = 0; l < 32; l++)
            if (cnt[l]) { if (l < mn) mn = l; if (l > mx) mx = l; }
        int code = 0;
        for (int l = mn; l <= mx; l++) {
            firstc[l] = code;
            code = (code + cnt[l]) << 1;
        }
    }
    __syncthreads();
    u32 tbits = 0;
    u64 dbits = 0;
    const u32 *hf = hfreq + ((size_t)b * MAX_TABLES + t) * HL;
    for (int j = threadIdx.x; j < alpha; j += 256) {
        int l = len[j];
        int before = 0;
        for (int i = 0; i < j; i++) before += (len[i] == l) ? 1 : 0;
        hcode[((size_t)b * MAX_TABLES + t) * HL + j] = ((u32)l << 24) | (u32)(firstc[l] + before);
        int prevl = (j == 0) ? l : len[j - 1];
        int d = l > prevl ? l - prevl : prevl - l;
        tbits += 2 * d + 1;
        dbits += (u64)hf[j] * (u64)l;
    }
    u32 tsum = block_reduce_add<256>(tbits, red);
    u32 dlo = block_reduce_add<256>((u32)(dbits & 0xffffffffu), red);
    u32 dhi = block_reduce_add<256>((u32)(dbits >> 32), red);
    // dlo may have carried: recombine exactly (sum of low words can exceed 32 bits only if
    // block sums do; block data bits < 2^25 so high words are all zero)
    if (threadIdx.x == 0) {
        tb[0] = 5 + tsum;
        tb[1] = ((u64)dhi << 32) + dlo;
    }
}

// ------------------------------------------------------------------ bit output helpers
// Stream bit p (0 = MSB of byte 0).  out words are the byte stream viewed as little-endian
// u32, so a big-endian word value is byte-swapped before it is ORed in.
__device__ __forceinline__ void put_bits(u32 *out, u64 bitpos, int len, u32 value) {
    if (len == 0) return;
    value &= (len >= 32) ? 0xffffffffu : ((1u << len) - 1u);
    u64 w = bitpos >> 5;
    int o = (int)(bitpos & 31);
    if (o + len <= 32) {
        atomicOr(&out[w], __builtin_bswap32(value << (32 - o - len)));
    } else {
        int lo = o + len - 32;
        atomicOr(&out[w], __builtin_bswap32(value >> lo));
        atomicOr(&out[w + 1], __builtin_bswap32(value << (32 - lo)));
    }
}

// selector MTF over table indices: list packed as 6 nibbles (nibble k = entry k)
#define SEL_IDENT 0x543210u
__device__ __forceinline__ u32 sel_mtf(u32 &list, u32 s) {
    u32 pos = 0;
#pragma unroll
    for (int k = 1; k < 6; k++)
        if (((list >> (4 * k)) & 15u) == s) pos = k;
    if (pos) {
        u32 lowmask = (1u << (4 * pos)) - 1u;
        u32 low = list & lowmask;
        u32 high = list & ~((1u << (4 * (pos + 1))) - 1u);
        list = high | (low << 4) | s;
    }
    return pos;
}
// state after a chunk from identity == recency order; composing = apply recency list of
// chunk B onto state A
__device__ __forceinline__ u32 sel_compose(u32 state, u32 rec, u32 seenmask) {
    // new = [entries of rec that were seen in chunk (front, in rec order)] ++ [state minus those]
    u32 out = 0;
    int o = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        u32 e = (rec >> (4 * k)) & 15u;
        if ((seenmask >> e) & 1u) { out |= e << (4 * o); o++; }
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {
        u32 e = (state >> (4 * k)) & 15u;
        if (!((seenmask >> e) & 1u)) { out |= e << (4 * o); o++; }
    }
    return out;
}

// ------------------------------------------------------------------ block plan
// one wave per block: selector MTF positions, section bit counts, total block bits.
// hdr[b] = {bits before selectors, selector bits, table bits, data bits}
// One workgroup of four waves per block: every thread takes a stretch of the selectors (a block has
// up to 18 002), the stretches' recency lists are composed along the wave (64 uniform steps from the
// identity; a thread keeps the list and the seen set in front of its stretch), the four waves'
// totals through LDS, and "state in front of the wave" composed with "what the wave's earlier threads
// did" is the list a thread replays its stretch from.  (One wave per block was 146 us of pure latency.)
__global__ __launch_bounds__(256) void k4_plan(const u32 *__restrict__ nsel,
                                               const u32 *__restrict__ ntables,
                                               const u32 *__restrict__ present,
                                               const u8 *__restrict__ sel, u8 *__restrict__ selpos,
                                               const u64 *__restrict__ tblbits,
                                               u64 *__restrict__ hdr) {
    const u32 b = blockIdx.x;
    const int lane = lane_id(), w = wave_id();
    const u32 ns = nsel[b];
    const u32 per = (ns + 255) / 256;
    const u32 s0 = min(ns, threadIdx.x * per), s1 = min(ns, s0 + per);
    const u8 *sp = sel + (size_t)b * SEL_STRIDE;
    u8 *op = selpos + (size_t)b * SEL_STRIDE;
    __shared__ u32 wlist[4], wseen[4], wbits[4];
    // phase A: the stretch's recency
    u32 list = SEL_IDENT, seen = 0;
    for (u32 i = s0; i < s1; i++) {
        u32 s = sp[i];
        sel_mtf(list, s);
        seen |= 1u << s;
    }
    // serial compose over the wave's lanes (64 steps, uniform), from the identity
    u32 state = SEL_IDENT, sacc = 0, erl = SEL_IDENT, ers = 0;
    for (int l = 0; l < 64; l++) {
        u32 rl = __builtin_amdgcn_readlane(list, l);
        u32 sm = __builtin_amdgcn_readlane(seen, l);
        if (lane == l) { erl = state; ers = sacc; }
        state = sel_compose(state, rl, sm);
        sacc |= sm;
    }
    if (lane == 0) { wlist[w] = state; wseen[w] = sacc; }
    __syncthreads();
    u32 wstart = SEL_IDENT;
    for (int q = 0; q < w; q++) wstart = sel_compose(wstart, wlist[q], wseen[q]);
    // phase B: replay
    list = sel_compose(wstart, erl, ers);
    u32 bits = 0;
    for (u32 i = s0; i < s1; i++) {
        u32 p = sel_mtf(list, sp[i]);
        op[i] = (u8)p;
        bits += p + 1;
    }
    bits = wave_reduce_add(bits);
    if (lane == 0) wbits[w] = bits;
    __syncthreads();
    if (threadIdx.x == 0) {
        const u32 selbits = wbits[0] + wbits[1] + wbits[2] + wbits[3];
        u32 used16 = 0;
        for (int i = 0; i < 16; i++) {
            u32 wv = present[b * 8 + (i >> 1)];
            if ((wv >> ((i & 1) * 16)) & 0xffffu) used16++;
        }
        u64 tb = 0, db = 0;
        const u32 nt = ntables[b];
        for (u32 t = 0; t < nt; t++) {
            tb += tblbits[((size_t)b * MAX_TABLES + t) * 2];
            db += tblbits[((size_t)b * MAX_TABLES + t) * 2 + 1];
        }
        hdr[b * 4 + 0] = 48 + 32 + 1 + 24 + 16 + 16 * used16 + 3 + 15;
        hdr[b * 4 + 1] = selbits;
        hdr[b * 4 + 2] = tb;
        hdr[b * 4 + 3] = db;
    }
}

// bit offsets of blocks: single workgroup (a scan over the blocks' bit totals).  If per_block
// (stage API) every block starts at b*stride_bits, else blocks abut starting at first_bit.
__global__ __launch_bounds__(256) void k4_offsets(const u64 *__restrict__ hdr, u64 *__restrict__ bitoff,
                                                  u32 nb, u64 first_bit, int per_block,
                                                  u64 stride_bits) {
    __shared__ u64 wsum[4];
    __shared__ u64 carry;
    const int lane = lane_id(), w = wave_id();
    if (threadIdx.x == 0) carry = first_bit;
    __syncthreads();
    for (u32 b0 = 0; b0 < nb; b0 += 256) {
        const u32 b = b0 + threadIdx.x;
        u64 tot = 0;
        if (b < nb) tot = hdr[b * 4] + hdr[b * 4 + 1] + hdr[b * 4 + 2] + hdr[b * 4 + 3];
        // wave scan in 32 bits: a block is < 2^25 bits, 64 of them fit
        const u32 inc = wave_incl_scan_add((u32)tot);
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        u64 base = carry;
        for (int q = 0; q < w; q++) base += wsum[q];
        if (b < nb) {
            bitoff[b] = per_block ? (u64)b * stride_bits : base + inc - tot;
            bitoff[nb + 1 + b] = tot;
        }
        __syncthreads();
        if (threadIdx.x == 255) carry = base + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) bitoff[nb] = per_block ? first_bit : carry;
}

// ------------------------------------------------------------------ header, selectors, tables
// one workgroup of 4 waves per block: thread 0 the fixed fields, all threads the selectors, wave
// w the coding tables w and w + 4 with the lanes over runs of consecutive symbols
__global__ __launch_bounds__(256) void k4_write_header(
    const u32 *__restrict__ nsel, const u32 *__restrict__ ntables, const u32 *__restrict__ alphaArr,
    const u32 *__restrict__ present, const u32 *__restrict__ crc, const u32 *__restrict__ orig,
    const u8 *__restrict__ selpos, const u8 *__restrict__ hlen, const u64 *__restrict__ hdr,
    const u64 *__restrict__ tblbits, const u64 *__restrict__ bitoff, u32 *__restrict__ out) {
    const u32 b = blockIdx.x;
    const int lane = lane_id(), w = wave_id();
    const u64 p0 = bitoff[b];
    const u32 ns = nsel[b];
    const u32 nt = ntables[b];
    const int alpha = (int)alphaArr[b];
    __shared__ u32 wsum[4];
    if (threadIdx.x == 0) {
        // S/BZip2BlockCompressor.swift:273-280, :89-144
        u64 p = p0;
        put_bits(out, p, 24, 0x314159); p += 24;
        put_bits(out, p, 24, 0x265359); p += 24;
        put_bits(out, p, 16, crc[b] >> 16); p += 16;
        put_bits(out, p, 16, crc[b] & 0xffff); p += 16;
        put_bits(out, p, 1, 0); p += 1;
        put_bits(out, p, 24, orig[b]); p += 24;
        u32 used = 0;
        for (int i = 0; i < 16; i++) {
            u32 wv = (present[b * 8 + (i >> 1)] >> ((i & 1) * 16)) & 0xffffu;
            if (wv) used |= 1u << (15 - i);
        }
        put_bits(out, p, 16, used); p += 16;
        for (int i = 0; i < 16; i++) {
            u32 wv = (present[b * 8 + (i >> 1)] >> ((i & 1) * 16)) & 0xffffu;
            if (wv) {
                // bit k (value i*16+k) is written k-th: reverse the 16 bits
                u32 rv = __brev(wv) >> 16;
                put_bits(out, p, 16, rv);
                p += 16;
            }
        }
        // S/BZip2HuffmanStageEncoder.swift:300-301
        put_bits(out, p, 3, nt); p += 3;
        put_bits(out, p, 15, ns); p += 15;
    }
    // selectors (:304-307): thread chunks, unary codes
    {
        const u32 per = (ns + 255) / 256;
        const u32 s0 = min(ns, threadIdx.x * per), s1 = min(ns, s0 + per);
        const u8 *sp = selpos + (size_t)b * SEL_STRIDE;
        u32 bits = 0;
        for (u32 i = s0; i < s1; i++) bits += sp[i] + 1;
        const u32 inc = wave_incl_scan_add(bits);
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        u32 base = 0;
        for (int q = 0; q < w; q++) base += wsum[q];
        u64 p = p0 + hdr[b * 4] + base + (inc - bits);
        for (u32 i = s0; i < s1; i++) {
            u32 v = sp[i];
            put_bits(out, p, (int)v + 1, ((1u << v) - 1u) << 1);
            p += v + 1;
        }
    }
    // tables (:310-327): start length in 5 bits, then per symbol the steps to its length and a 0
    for (u32 t = (u32)w; t < nt; t += 4) {
        u64 p = p0 + hdr[b * 4] + hdr[b * 4 + 1];
        for (u32 q = 0; q < t; q++) p += tblbits[((size_t)b * MAX_TABLES + q) * 2];
        const u8 *l = hlen + ((size_t)b * MAX_TABLES + t) * HL;
        const int per = (alpha + 63) / 64;
        const int j0 = min(alpha, lane * per), j1 = min(alpha, j0 + per);
        const int before = l[j0 > 0 ? j0 - 1 : 0];
        u32 bits = 0;
        int cur = before;
        for (int j = j0; j < j1; j++) {
            const int len = l[j];
            bits += 2u * (u32)(len > cur ? len - cur : cur - len) + 1u;
            cur = len;
        }
        const u32 inc = wave_incl_scan_add(bits);
        if (lane == 0) put_bits(out, p, 5, (u32)l[0]);
        p += 5 + (inc - bits);
        cur = before;
        for (int j = j0; j < j1; j++) {
            const int len = l[j];
            const u32 v = (cur < len) ? 2u : 3u;
            int d = len > cur ? len - cur : cur - len;
            while (d-- > 0) { put_bits(out, p, 2, v); p += 2; }
            put_bits(out, p, 1, 0); p += 1;
            cur = len;
        }
    }
}

// ------------------------------------------------------------------ symbol packing (:336-353)
// phase 1: bits per tile of PK_TILE symbols
__global__ __launch_bounds__(256) void k7_pack_p1(const u16 *__restrict__ mtf,
                                                  const u32 *__restrict__ mtflen,
                                                  const u8 *__restrict__ sel,
                                                  const u8 *__restrict__ hlen,
                                                  u32 *__restrict__ pkagg, u32 S2, u32 PKT) {
    const u32 b = blockIdx.y;
    const u32 L = mtflen[b];
    const u32 base = blockIdx.x * PK_TILE;
    if (base >= L) return;
    __shared__ u8 ll[MAX_TABLES * HL];
    __shared__ u32 red[8];
    for (u32 q = threadIdx.x; q < MAX_TABLES * HL; q += 256) ll[q] = hlen[(size_t)b * MAX_TABLES * HL + q];
    __syncthreads();
    const u16 *src = mtf + (size_t)b * S2;
    const u8 *sp = sel + (size_t)b * SEL_STRIDE;
    u32 bits = 0;
    // clamped loads first (all in flight), then the LDS lookups
    u16 sv[PK_TILE / 256];
    u8 tv[PK_TILE / 256];
#pragma unroll
    for (int k = 0; k < PK_TILE / 256; k++) {
        const u32 j = base + k * 256 + threadIdx.x;
        const u32 jc = j < L ? j : L - 1;
        sv[k] = src[jc];
        tv[k] = sp[jc / GROUP];
    }
#pragma unroll
    for (int k = 0; k < PK_TILE / 256; k++) {
        u32 j = base + k * 256 + threadIdx.x;
        if (j < L) bits += ll[tv[k] * HL + sv[k]];
    }
    u32 tot = block_reduce_add<256>(bits, red);
    if (threadIdx.x == 0) pkagg[(size_t)b * PKT + blockIdx.x] = tot;
}

// phase 2: exclusive scan per block (one workgroup per block)
__global__ __launch_bounds__(256) void k7_pack_p2(const u32 *__restrict__ mtflen,
                                                  u32 *__restrict__ pkagg, u32 PKT) {
    const u32 b = blockIdx.x;
    const u32 tiles = (mtflen[b] + PK_TILE - 1) / PK_TILE;
    __shared__ u32 sm[8];
    u32 carry = 0;
    for (u32 t0 = 0; t0 < tiles; t0 += 256) {
        u32 t = t0 + threadIdx.x;
        u32 v = (t < tiles) ? pkagg[(size_t)b * PKT + t] : 0;
        u32 tot;
        u32 ex = block_excl_scan_add<256>(v, sm, &tot);
        if (t < tiles) pkagg[(size_t)b * PKT + t] = carry + ex;
        carry += tot;
        __syncthreads();
    }
}

// phase 3: assemble the tile's bits in LDS, flush words (edges by atomicOr)
#define PK_PER_THREAD (PK_TILE / 256)
#define PK_WORDS (PK_TILE * 20 / 32 + 4)
__global__ __launch_bounds__(256) void k7_pack_p3(
    const u16 *__restrict__ mtf, const u32 *__restrict__ mtflen, const u8 *__restrict__ sel,
    const u32 *__restrict__ hcode, const u32 *__restrict__ pkagg, const u64 *__restrict__ hdr,
    const u64 *__restrict__ bitoff, u32 *__restrict__ out, u32 S2, u32 PKT) {
    const u32 b = blockIdx.y;
    const u32 L = mtflen[b];
    const u32 base = blockIdx.x * PK_TILE;
    if (base >= L) return;
    __shared__ u32 codes[MAX_TABLES * HL];
    __shared__ u32 words[PK_WORDS];
    __shared__ u32 sm[8];
    for (u32 q = threadIdx.x; q < MAX_TABLES * HL; q += 256) codes[q] = hcode[(size_t)b * MAX_TABLES * HL + q];
    for (u32 q = threadIdx.x; q < PK_WORDS; q += 256) words[q] = 0;
    __syncthreads();
    const u16 *src = mtf + (size_t)b * S2;
    const u8 *sp = sel + (size_t)b * SEL_STRIDE;
    u32 mc[PK_PER_THREAD];
    u32 bits = 0;
    u16 sv[PK_PER_THREAD];
    u8 tv[PK_PER_THREAD];
#pragma unroll
    for (int k = 0; k < PK_PER_THREAD; k++) {
        const u32 j = base + threadIdx.x * PK_PER_THREAD + k;
        const u32 jc = j < L ? j : L - 1;
        sv[k] = src[jc];
        tv[k] = sp[jc / GROUP];
    }
#pragma unroll
    for (int k = 0; k < PK_PER_THREAD; k++) {
        u32 j = base + threadIdx.x * PK_PER_THREAD + k;
        u32 m = 0;
        if (j < L) m = codes[tv[k] * HL + sv[k]];
        mc[k] = m;
        bits += m >> 24;
    }
    u32 tot;
    u32 ex = block_excl_scan_add<256>(bits, sm, &tot);
    const u64 tile_bit = bitoff[b] + hdr[b * 4] + hdr[b * 4 + 1] + hdr[b * 4 + 2] +
                         pkagg[(size_t)b * PKT + blockIdx.x];
    const u32 skew = (u32)(tile_bit & 31);
    u32 p = skew + ex;
#pragma unroll
    for (int k = 0; k < PK_PER_THREAD; k++) {
        u32 len = mc[k] >> 24;
        if (len) {
            u32 value = mc[k] & ((1u << len) - 1u);
            u32 w = p >> 5;
            u32 o = p & 31;
            if (o + len <= 32) {
                atomicOr(&words[w], value << (32 - o - len));
            } else {
                u32 lo = o + len - 32;
                atomicOr(&words[w], value >> lo);
                atomicOr(&words[w + 1], value << (32 - lo));
            }
            p += len;
        }
    }
    __syncthreads();
    const u32 nwords = (skew + tot + 31) >> 5;
    u32 *gw = out + (tile_bit >> 5);
    for (u32 q = threadIdx.x; q < nwords; q += 256) {
        u32 v = __builtin_bswap32(words[q]);
        if (q == 0 || q == nwords - 1) {
            if (v) atomicOr(&gw[q], v);
        } else {
            gw[q] = v;
        }
    }
}

// ------------------------------------------------------------------ stream header / trailer
// S/BZip2OutputStream.swift:139,154-157,181-184.  ctl_out[0..1] = end bit (lo, hi), [2] = crc
// One wave.  Lane 0 frames the stream; all lanes add up the batch's statistics (what the host used to
// fetch as three arrays): ctl_out[4..12) = sum of rounds, of block lengths, of length x rounds (u64
// each) and the number of periodic blocks.
__global__ void k8_frame(const u32 *__restrict__ crc, const u64 *__restrict__ bitoff, u32 nb,
                         u32 *__restrict__ out, int level, int write_header, int write_trailer,
                         u32 crc_in, u32 *__restrict__ ctl_out, const u32 *__restrict__ rounds,
                         const u32 *__restrict__ periodic, const u32 *__restrict__ nArr) {
    if (blockIdx.x != 0) return;
    {
        __shared__ unsigned long long acc[4];
        if (threadIdx.x < 4) acc[threadIdx.x] = 0;
        __syncthreads();
        unsigned long long sr = 0, sn = 0, snr = 0, sp = 0;
        for (u32 b = threadIdx.x; b < nb; b += blockDim.x) {
            sr += rounds[b];
            sn += nArr[b];
            snr += (unsigned long long)nArr[b] * rounds[b];
            sp += periodic[b] > 1;
        }
        atomicAdd(&acc[0], sr);
        atomicAdd(&acc[1], sn);
        atomicAdd(&acc[2], snr);
        atomicAdd(&acc[3], sp);
        __syncthreads();
        if (threadIdx.x < 4) {
            ctl_out[4 + 2 * threadIdx.x] = (u32)(acc[threadIdx.x] & 0xffffffffu);
            ctl_out[5 + 2 * threadIdx.x] = (u32)(acc[threadIdx.x] >> 32);
        }
    }
    if (threadIdx.x != 0) return;
    if (write_header) {
        put_bits(out, 0, 16, 0x425a);
        put_bits(out, 16, 8, 0x68);
        put_bits(out, 24, 8, (u32)('0' + level));
    }
    u32 sc = crc_in;
    for (u32 b = 0; b < nb; b++) sc = ((sc << 1) | (sc >> 31)) ^ crc[b];
    u64 p = bitoff[nb];
    if (write_trailer) {
        put_bits(out, p, 24, 0x177245); p += 24;
        put_bits(out, p, 24, 0x385090); p += 24;
        put_bits(out, p, 16, sc >> 16); p += 16;
        put_bits(out, p, 16, sc & 0xffff); p += 16;
    }
    ctl_out[0] = (u32)(p & 0xffffffffu);
    ctl_out[1] = (u32)(p >> 32);
    ctl_out[2] = sc;
}

// ------------------------------------------------------------------ segment placement (K8 across shards)
// Copies nbits bits of src (bit 0 = MSB of byte 0; 4-byte aligned) to stream bit `dst` of out: one
// 32-bit destination word per thread, two source words funnel-shifted.  Words shared with the
// neighbouring segments are OR-ed in (out is zeroed before), inner words are plain stores.
__global__ __launch_bounds__(256) void k8_place(const u32 *__restrict__ src, u64 nbits,
                                                u32 *__restrict__ out, u64 dst) {
    const u64 w0 = dst >> 5, w1 = (dst + nbits - 1) >> 5;
    const u64 j = (u64)blockIdx.x * 256u + threadIdx.x; // destination word w0 + j
    const u64 w = w0 + j;
    if (w > w1) return;
    const int sh = (int)(dst & 31);
    const u64 nsrcw = (nbits + 31) >> 5;
    const u32 a = (sh && j > 0) ? __builtin_bswap32(src[j - 1]) : 0u;
    const u32 b = (j < nsrcw) ? __builtin_bswap32(src[j]) : 0u;
    u32 v = sh ? ((a << (32 - sh)) | (b >> sh)) : b;
    // keep the stream bits [lo, hi) of this word
    const u64 lo = (w << 5) > dst ? (w << 5) : dst;
    const u64 end = dst + nbits;
    const u64 hi = ((w + 1) << 5) < end ? ((w + 1) << 5) : end;
    const u32 mask = (0xffffffffu >> (u32)(lo & 31)) & (0xffffffffu << (31u - (u32)((hi - 1) & 31)));
    v &= mask;
    if (w == w0 || w == w1) {
        if (v) atomicOr(&out[w], __builtin_bswap32(v));
    } else {
        out[w] = __builtin_bswap32(v);
    }
}

// stream header at bit 0, end marker + stream CRC at end_bit (S/BZip2OutputStream.swift:154-157,181-184)
__global__ void k8_ends(u32 *__restrict__ out, int level, u64 end_bit, u32 scrc) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    put_bits(out, 0, 16, 0x425a);
    put_bits(out, 16, 8, 0x68);
    put_bits(out, 24, 8, (u32)('0' + level));
    u64 p = end_bit;
    put_bits(out, p, 24, 0x177245); p += 24;
    put_bits(out, p, 24, 0x385090); p += 24;
    put_bits(out, p, 16, scrc >> 16); p += 16;
    put_bits(out, p, 16, scrc & 0xffff);
}

void run_place(bz2hip_ctx *c, const u8 *d_src, u64 nbits, u8 *d_out, u64 dst_bit) {
    const u64 words = ((dst_bit + nbits - 1) >> 5) - (dst_bit >> 5) + 1;
    ProfScope ps(c, P_FINAL);
    k8_place<<<(unsigned)((words + 255) / 256), 256, 0, c->stream>>>((const u32 *)d_src, nbits, (u32 *)d_out, dst_bit);
    KCHECK();
}

void run_ends(bz2hip_ctx *c, u8 *d_out, int level, u64 end_bit, u32 scrc) {
    ProfScope ps(c, P_FINAL);
    k8_ends<<<1, 64, 0, c->stream>>>((u32 *)d_out, level, end_bit, scrc);
    KCHECK();
}

// ------------------------------------------------------------------ drivers
void run_huffman(bz2hip_ctx *c, int nb, u32 maxmtf) {
    Batch &B = c->batch;
    if (nb == 0) return;
    {
        ProfScope ps(c, P_HUF_SEED);
        k4_seed<<<nb, 256, 0, c->stream>>>(B.mtflen, B.alpha, B.freq, B.ntables, B.nsel, B.hlen, nb);
        KCHECK();
    }
    const u32 ngroups = (maxmtf + GROUP - 1) / GROUP;
    const u32 parts = (ngroups + HS_GROUPS_PER_WG - 1) / HS_GROUPS_PER_WG;
    for (int pass = 3; pass >= 0; pass--) {
        {
            ProfScope ps(c, P_HUF_SELECT);
            k4_select<<<dim3(parts, nb), 256, 0, c->stream>>>(B.mtf, B.mtflen, B.ntables, B.hlen,
                                                              B.hpart, B.sel, B.S + 64, pass == 0);
            KCHECK();
        }
        {
            ProfScope ps(c, P_HUF_LENGTHS);
            if ((size_t)MAX_TABLES * nb > 2048) // more tables than 256-thread workgroups fit on the chip at once
                k4_lengths<64><<<dim3(MAX_TABLES, nb), 64, 0, c->stream>>>(B.mtflen, B.alpha, B.ntables,
                                                                           B.hpart, B.hlen, B.hfreq);
            else
                k4_lengths<256><<<dim3(MAX_TABLES, nb), 256, 0, c->stream>>>(B.mtflen, B.alpha, B.ntables,
                                                                             B.hpart, B.hlen, B.hfreq);
            KCHECK();
        }
    }
    {
        ProfScope ps(c, P_HUF_CODES);
        k4_codes<<<dim3(MAX_TABLES, nb), 256, 0, c->stream>>>(B.alpha, B.ntables, B.hlen, B.hfreq,
                                                              B.hcode, B.tblbits);
        KCHECK();
    }
}

void run_pack(bz2hip_ctx *c, int nb, u32 maxmtf, u8 *d_out, u64 first_bit, bool per_block,
              size_t bits_stride, size_t out_cap_bytes, std::vector<u64> *h_nbits) {
    Batch &B = c->batch;
    if (nb == 0) return;
    run_tie_finish(c, nb); // the block headers carry the origin pointers: they must be final now
    const u32 PKT = (B.S + 64 + PK_TILE - 1) / PK_TILE;
    {
        ProfScope ps(c, P_HUF_HEADER);
        k4_plan<<<nb, 256, 0, c->stream>>>(B.nsel, B.ntables, B.present, B.sel, B.selpos, B.tblbits,
                                          B.hdrbits);
        KCHECK();
        k4_offsets<<<1, 256, 0, c->stream>>>(B.hdrbits, B.bitoff, nb, first_bit, per_block ? 1 : 0,
                                            (u64)bits_stride * 8);
        KCHECK();
        // exact size check before a single bit is written
        std::vector<u64> hbv;
        u64 *hb = (u64 *)c->h_pin; // page-locked when it fits (it does up to 8191 blocks a batch)
        if ((size_t)4 * (nb + 1) > H_PIN_WORDS) {
            hbv.resize(2 * (nb + 1));
            hb = hbv.data();
        }
        HIPCHK(hipMemcpyAsync(hb, B.bitoff, sizeof(u64) * 2 * (nb + 1), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (per_block) {
            for (int b = 0; b < nb; b++)
                if ((hb[nb + 1 + b] + 7) / 8 + 8 > bits_stride)
                    throw HipError(BZ2HIP_E_CAP, "bits_stride too small for a compressed block");
        } else {
            // trailer (80 bits) + word-granular stores need slack
            if ((hb[nb] + 80 + 7) / 8 + 8 > out_cap_bytes)
                throw HipError(BZ2HIP_E_CAP, "output buffer too small");
        }
        if (h_nbits) h_nbits->assign(hb + nb + 1, hb + nb + 1 + nb);
        k4_write_header<<<nb, 256, 0, c->stream>>>(B.nsel, B.ntables, B.alpha, B.present, B.crc,
                                                   B.orig, B.selpos, B.hlen, B.hdrbits, B.tblbits,
                                                   B.bitoff, (u32 *)d_out);
        KCHECK();
    }
    {
        ProfScope ps(c, P_PACK);
        const u32 tiles = (maxmtf + PK_TILE - 1) / PK_TILE;
        dim3 grid(tiles, nb);
        k7_pack_p1<<<grid, 256, 0, c->stream>>>(B.mtf, B.mtflen, B.sel, B.hlen, B.pkagg, B.S + 64, PKT);
        KCHECK();
        k7_pack_p2<<<nb, 256, 0, c->stream>>>(B.mtflen, B.pkagg, PKT);
        KCHECK();
        k7_pack_p3<<<grid, 256, 0, c->stream>>>(B.mtf, B.mtflen, B.sel, B.hcode, B.pkagg, B.hdrbits,
                                                B.bitoff, (u32 *)d_out, B.S + 64, PKT);
        KCHECK();
    }
}

void run_frame(bz2hip_ctx *c, int nb, u8 *d_out, int level, bool header, bool trailer, u32 crc_in,
               u64 *end_bit, u32 *crc_out, u64 *stats) {
    Batch &B = c->batch;
    ProfScope ps(c, P_FINAL);
    k8_frame<<<1, 64, 0, c->stream>>>(B.crc, B.bitoff, nb, (u32 *)d_out, level, header ? 1 : 0,
                                      trailer ? 1 : 0, crc_in, B.ctl + 48, B.rounds, B.periodic, B.n);
    KCHECK();
    u32 *h = c->h_ctl + 128; // page-locked (the first 64 words belong to the sort's round loop)
    HIPCHK(hipMemcpyAsync(h, B.ctl + 48, 12 * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *end_bit = ((u64)h[1] << 32) | h[0];
    *crc_out = h[2];
    if (stats)
        for (int k = 0; k < 4; k++) stats[k] = ((u64)h[5 + 2 * k] << 32) | h[4 + 2 * k];
}
