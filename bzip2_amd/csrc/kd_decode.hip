// kd_decode.hip -- block-parallel bzip2 decoder.
//
// Replaces BZip2InputStream (S/BZip2InputStream.swift:154-242), BZip2BlockDecompressor
// (S/BZip2BlockDecompressor.swift:199-483), BZip2HuffmanStageDecoder (S/BZip2HuffmanStageDecoder.swift:
// 84-181) and BZip2BitInputStream (S/BZip2BitInputStream.swift:53-146) for a whole stream resident in
// HBM.  The reference decodes one block after the other; here every block is decoded at once:
//   kd_scan_magic : every bit offset is tested for the 48-bit block magic (blocks abut at any bit)
//   kd_huff       : two wavefronts per candidate: header, selectors, code lengths, canonical tables,
//                   then wave 0 decodes Huffman symbols into an LDS ring (every lane looks up the code
//                   at its own bit offset of a 128-bit window, a scalar chain follows the real codes)
//                   and wave 1 turns batches of 64 symbols into MTF positions, one per output byte
//                   (RUNA-RUNB runs become runs of position 0), with the reference's size checks;
//                   also origin pointer, stored CRC, end bit
//   kd_imtf_*     : MTF positions -> BWT bytes for all chunks of all blocks at once (a chunk's effect
//                   on the list is a permutation; permutations are composed per block)
//   host          : walks the chain magic -> end bit -> magic from bit 32; anything a candidate
//                   that is not on the chain decoded is ignored
//   kd_hist / kd_merge : stable counting sort of the BWT bytes = the reference's merged pointers
//   kd_walk_a/b/c : the n-step pointer walk as list ranking: every 128th node (+ the origin) is a
//                   ruler; a) each ruler walks to the next ruler, b) one lane per block walks the
//                   ruler chain from the origin and gives each ruler its output offset, c) each
//                   ruler walks again and writes its bytes.  A walk that returns to the origin before
//                   n steps (block = w^k: k cycles) is written once and replicated (kd_cycfill).
//   kd_unrle_*    : final run-length decoding over rows of 64 bytes: which bytes are repeat counts
//                   is a 2-state automaton over runs of equal bytes (state = "the next byte is a
//                   count"), scanned inside the row by function composition; a block is cut at run
//                   heads into 16 pieces, each counted for both entry states, resolved, then written
//   k1_crc_p1/p2  : block CRCs of the decoded bytes (the encoder's kernels)
// Randomised blocks (S/BZip2BlockDecompressor.swift:367-374) are not supported: BZ2HIP_E_RANDOMISED.
#include "ctx.hpp"
#include <algorithm>
#include <vector>

#define KD_WIN 4096    // compressed bytes staged in LDS per refill
#define KD_RULER 128
#define KD_MAXLEN 23   // HUFFMAN_DECODE_MAXIMUM_CODE_LENGTH, S/BZip2Constants.swift:57

enum { KS_OK = 0, KS_INSUFFICIENT, KS_TABLES, KS_BLOCK_SIZE, KS_DECODE, KS_RANDOMISED };
// meta words per candidate
enum { KM_STATUS = 0, KM_N, KM_ORIG, KM_CRC, KM_END_LO, KM_END_HI, KM_WORDS = 8 };

void run_crc_blocks(bz2hip_ctx *c, const u8 *d_data, const u32 *d_starts, int nb, u32 max_len,
                    u32 *d_cagg, u32 ETPB, u32 *d_crc); // k1_rle1.hip

// ------------------------------------------------------------------ block magic scan
__global__ __launch_bounds__(256) void kd_scan_magic(const u8 *__restrict__ in, u64 nbytes,
                                                     u64 *__restrict__ cand, u32 *__restrict__ ncand,
                                                     u32 cap) {
    const u64 p = (u64)blockIdx.x * 256u + threadIdx.x;
    if (p >= nbytes) return;
    u64 v = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) v = (v << 8) | (u64)((p + q < nbytes) ? in[p + q] : 0);
#pragma unroll
    for (int s = 0; s < 8; s++) {
        if (((v << s) >> 16) == 0x314159265359ull && p * 8u + s + 48u <= nbytes * 8u) {
            const u32 slot = atomicAdd(ncand, 1u);
            if (slot < cap) cand[slot] = p * 8u + s;
        }
    }
}

// ------------------------------------------------------------------ Huffman + MTF stage
// MSB-first bit reader over an LDS window of the stream, refilled 32 bits at a time (window words
// are aligned to stream offsets that are multiples of 4, whatever the pointer's alignment)
struct KdBits {
    const u8 *in;
    u64 nbytes, wpos, wbase; // wpos = next stream word (4 bytes) to enter the buffer; window = [wbase, wbase + KD_WIN/4)
    u64 hi, lo;              // 128-bit buffer, MSB of hi first
    int bc;                  // valid bits
    u32 *win;
    __device__ __forceinline__ void load_window(u64 atw) {
        wbase = atw;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        u8 *wb = (u8 *)win;
        for (int j = 0; j < KD_WIN / 64; j++) {
            const u64 a = atw * 4u + (u64)j * 64u + (u64)lane_id();
            wb[j * 64 + lane_id()] = (a < nbytes) ? in[a] : (u8)0;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ void fill() { // at least 97 valid bits afterwards
        while (bc <= 96) {
            if (wpos >= wbase + KD_WIN / 4) load_window(wpos);
            // (readfirstlane: the reader's state is wave-uniform, this keeps it in scalar registers)
            const u32 w = __builtin_amdgcn_readfirstlane(__builtin_bswap32(win[wpos - wbase]));
            const unsigned __int128 t = (unsigned __int128)w << (96 - bc);
            hi |= (u64)(t >> 64);
            lo |= (u64)t;
            bc += 32;
            wpos++;
        }
    }
    __device__ __forceinline__ void start(u64 bit) {
        hi = 0; lo = 0; bc = 0;
        wpos = bit >> 5;
        load_window(wpos);
        fill();
        skip((int)(bit & 31u));
    }
    // The reader's state is the same in every lane, but the compiler cannot always prove it (a
    // value that meets a lane-dependent branch anywhere is kept in vector registers, and loops over
    // it then run with exec masks and VALU compares, ~25 instructions per step instead of ~10):
    // pin it to scalar registers.
    __device__ __forceinline__ static u64 uni64(u64 v) {
        return ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(v >> 32)) << 32) |
               (u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)v);
    }
    __device__ __forceinline__ void pin() {
        hi = uni64(hi); lo = uni64(lo); wpos = uni64(wpos); wbase = uni64(wbase);
        bc = __builtin_amdgcn_readfirstlane(bc);
    }
    __device__ __forceinline__ u32 peek(int k) { return k ? (u32)(hi >> (64 - k)) : 0u; } // k <= 32
    __device__ __forceinline__ void skip(int k) { // k <= 127
        const unsigned __int128 t = (((unsigned __int128)hi << 64) | lo) << k;
        hi = (u64)(t >> 64);
        lo = (u64)t;
        bc -= k;
    }
    __device__ __forceinline__ u32 bits(int k) { fill(); u32 v = peek(k); skip(k); return v; }
    __device__ __forceinline__ u64 consumed() const { return wpos * 32u - (u64)bc; }
    // the reference fails when it needs a byte that is not there: a consumed bit past the end
    __device__ __forceinline__ bool over() const { return consumed() > nbytes * 8u; }
};

#define KD_LUT_BITS 10

#define KD_RING 2048        // symbols between the two waves of kd_huff
#define KD_SPIN_MAX (1 << 22)
#define KD_SYM_ERR 0x8000u  // ring entry: the producer stopped with status (entry & 0xff)
// The ring, its head and its tail live in LDS, and the LDS serves one wave's operations in order:
// "ring entries, then head" (and "ring reads, then tail") needs no more than that the compiler
// keeps the order and that the wave's own LDS operations have been issued.  An acquire / release
// at workgroup scope would also wait for the wave's outstanding GLOBAL accesses (the consumer's row
// stores: a memory round trip per batch of 64 symbols).
#define KD_LDS_ORDER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// The chain through one register of window entries (lane = bit offset), 11 instructions per symbol
// (the compiler makes 16 of the same loop, with the conditions turned into mask arithmetic):
//   do { e = readlane(ment, off); if (e & 0x8000) break; visited |= 1 << off;
//        off = (e >> 16) & 127; n++; } while (!(e & 0x800000) && n < lim);
// Entry: bit 15 = stop before it (no entry / no symbol / end of block), bits 16..22 = offset of the
// next code, bit 23 = that offset lies outside the register.  Returns the last entry read.
__device__ __forceinline__ u32 kd_chain(u32 ment, u32 &off, u32 &n, u32 lim, u64 &visited) {
    u32 e;
    u64 t64;
    asm volatile("1:\n\t"
                 "v_readlane_b32 %[e], %[ment], %[off]\n\t"
                 "s_bitcmp1_b32 %[e], 15\n\t"
                 "s_cbranch_scc1 2f\n\t"
                 "s_lshl_b64 %[t], 1, %[off]\n\t"
                 "s_or_b64 %[vis], %[vis], %[t]\n\t"
                 "s_bfe_u32 %[off], %[e], 0x70010\n\t"
                 "s_add_u32 %[n], %[n], 1\n\t"
                 "s_bitcmp1_b32 %[e], 23\n\t"
                 "s_cbranch_scc1 2f\n\t"
                 "s_cmp_lt_u32 %[n], %[lim]\n\t"
                 "s_cbranch_scc1 1b\n\t"
                 "2:"
                 : [e] "=&s"(e), [off] "+s"(off), [vis] "+s"(visited), [n] "+s"(n), [t] "=&s"(t64)
                 : [ment] "v"(ment), [lim] "s"(lim)
                 : "scc");
    return e;
}

// The same chain, two symbols per trip while neither entry carries a flag and two more fit under
// lim: 17 instructions per pair instead of 22 (one flag test for "stop before" and "leaves the
// register" of the first entry, one visited update, one count, one limit test).  Anything else --
// a flagged entry, the last symbol before lim -- goes through the single-symbol loop above, which
// this block ends with.  (A lane select written by a scalar instruction needs four issue slots
// before v_readlane uses it: the instruction order keeps them, and the block starts with a nop
// because it cannot see what the compiler put before it.)
__device__ __forceinline__ u32 kd_chain2(u32 ment, u32 &off, u32 &n, u32 lim, u64 &visited) {
    if (lim < 2u + n) return kd_chain(ment, off, n, lim, visited);
    u32 e, e1, o2, tt;
    u64 ta, tb;
    const u32 lim2 = lim - 2u;
    asm volatile("s_nop 3\n\t"
                 "1:\n\t"
                 "v_readlane_b32 %[e1], %[ment], %[off]\n\t"
                 "s_bfe_u32 %[o2], %[e1], 0x70010\n\t"
                 "s_lshl_b64 %[ta], 1, %[off]\n\t"
                 "s_and_b32 %[tt], %[e1], 0x808000\n\t"
                 "s_cbranch_scc1 3f\n\t"
                 "s_lshl_b64 %[tb], 1, %[o2]\n\t"
                 "v_readlane_b32 %[e], %[ment], %[o2]\n\t"
                 "s_bitcmp1_b32 %[e], 15\n\t"
                 "s_cbranch_scc1 3f\n\t"
                 "s_or_b64 %[ta], %[ta], %[tb]\n\t"
                 "s_or_b64 %[vis], %[vis], %[ta]\n\t"
                 "s_bfe_u32 %[off], %[e], 0x70010\n\t"
                 "s_add_u32 %[n], %[n], 2\n\t"
                 "s_bitcmp1_b32 %[e], 23\n\t"
                 "s_cbranch_scc1 2f\n\t"
                 "s_cmp_le_u32 %[n], %[lim2]\n\t"
                 "s_cbranch_scc1 1b\n\t"
                 "s_cmp_lt_u32 %[n], %[lim]\n\t"
                 "s_cbranch_scc0 2f\n\t"
                 "3:\n\t"
                 "v_readlane_b32 %[e], %[ment], %[off]\n\t"
                 "s_bitcmp1_b32 %[e], 15\n\t"
                 "s_cbranch_scc1 2f\n\t"
                 "s_lshl_b64 %[ta], 1, %[off]\n\t"
                 "s_or_b64 %[vis], %[vis], %[ta]\n\t"
                 "s_bfe_u32 %[off], %[e], 0x70010\n\t"
                 "s_add_u32 %[n], %[n], 1\n\t"
                 "s_bitcmp1_b32 %[e], 23\n\t"
                 "s_cbranch_scc1 2f\n\t"
                 "s_cmp_lt_u32 %[n], %[lim]\n\t"
                 "s_cbranch_scc1 3b\n\t"
                 "2:"
                 : [e] "=&s"(e), [e1] "=&s"(e1), [o2] "=&s"(o2), [tt] "=&s"(tt), [ta] "=&s"(ta), [tb] "=&s"(tb),
                   [off] "+s"(off), [vis] "+s"(visited), [n] "+s"(n)
                 : [ment] "v"(ment), [lim] "s"(lim), [lim2] "s"(lim2)
                 : "scc");
    return e;
}

// Two waves per block.  Wave 0 parses the header and builds the tables, then only decodes Huffman
// symbols into an LDS ring; wave 1 turns symbols into the MTF position of every output byte (runs
// expanded) and applies the reference's size checks.  The two run on different SIMDs of the CU.
// Errors travel through the ring in symbol order, so the first one in the reference's order wins; a
// wave that gives up raises s_abort for the other.
__global__ __launch_bounds__(128) void kd_huff(const u8 *__restrict__ in, u64 nbytes,
                                               const u64 *__restrict__ cand, u32 block_size, u32 S,
                                               u8 *__restrict__ bwt, u32 *__restrict__ meta,
                                               u8 *__restrict__ symmap_out,
                                               const u32 *__restrict__ list) {
    const u32 b = list[blockIdx.x]; // candidate slot of the batch
    const int lane = lane_id();
    const int wv = wave_id();
    __shared__ u32 win[KD_WIN / 4];
    __shared__ u16 lut[6][1 << KD_LUT_BITS]; // next 10 bits -> (symbol << 5) | length; 0 = walk the limits
    __shared__ u8 sel[SEL_MAX + 2];
    __shared__ u8 lens[6][MAX_ALPHA + 2];
    __shared__ int limits[6][KD_MAXLEN + 1];
    __shared__ int bases[6][KD_MAXLEN + 2];
    __shared__ u16 symbols[6][MAX_ALPHA + 2];
    __shared__ int minlen[6];
    __shared__ __attribute__((aligned(16))) u8 symmap[256];
    __shared__ u8 tmtf[256];
    __shared__ u16 ring[KD_RING];
    __shared__ u32 s_head, s_tail, s_abort;
    __shared__ u32 s_status, s_eob, s_nsel, s_orig, s_crc, s_len, s_end_lo, s_end_hi;
    u32 *mt = meta + (size_t)b * KM_WORDS;
    u8 *out = bwt + (size_t)b * S;

    KdBits br;
    br.in = in; br.nbytes = nbytes; br.win = win;
    br.hi = 0; br.lo = 0; br.bc = 0; br.wpos = 0; br.wbase = 0;
    if (threadIdx.x == 0) { s_head = 0; s_tail = 0; s_abort = 0; s_len = 0; s_end_lo = 0; s_end_hi = 0; }
    if (wv == 0) {
        br.start(cand[b] + 48u);
        u32 status = KS_OK, orig = 0, crc = 0;
        // zeroed tables: a selector may name a table the block does not define (the reference's
        // arrays are zero-initialised, S/BZip2HuffmanStageDecoder.swift:44-61)
        for (int q = lane; q < 6 * (KD_MAXLEN + 1); q += 64) (&limits[0][0])[q] = 0;
        for (int q = lane; q < 6 * (KD_MAXLEN + 2); q += 64) (&bases[0][0])[q] = 0;
        for (int q = lane; q < 6 * (MAX_ALPHA + 2); q += 64) { (&symbols[0][0])[q] = 0; (&lens[0][0])[q] = 0; }
        for (int q = lane; q < 256; q += 64) { symmap[q] = 0; tmtf[q] = (u8)q; }
        if (lane < 6) minlen[lane] = 0;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        int eob = 0, ntab = 0, nsel = 0;
        do {
            // block header, S/BZip2BlockDecompressor.swift:474-477
            crc = br.bits(16) << 16;
            crc |= br.bits(16);
            const u32 randomised = br.bits(1);
            orig = br.bits(24);
            if (br.over()) { status = KS_INSUFFICIENT; break; }
            if (randomised) { status = KS_RANDOMISED; break; }
            // symbol map, :203-221
            int nsym = 0;
            const u32 ranges = br.bits(16);
            for (int i = 0; i < 16; i++) {
                if (ranges & (0x8000u >> i)) {
                    const u32 m = br.bits(16);
                    for (int j = 0; j < 16; j++)
                        if (m & (0x8000u >> j)) symmap[nsym++] = (u8)(i * 16 + j);
                }
            }
            if (br.over()) { status = KS_INSUFFICIENT; break; }
            eob = nsym + 1;
            ntab = (int)br.bits(3);
            nsel = (int)br.bits(15);
            if (br.over()) { status = KS_INSUFFICIENT; break; }
            if (ntab < 2 || ntab > 6 || nsel < 1 || nsel > SEL_MAX - 1) { status = KS_TABLES; break; }
            // selectors: unary, MTF over the table indices, :238-243
            for (int i = 0; i < nsel && status == KS_OK; i++) {
                int u = 0;
                for (;;) {
                    br.fill();
                    // leading ones of the first 64 bits of the buffer (at least 97 are valid)
                    const int ones = (~br.hi) ? __clzll((long long)~br.hi) : 64;
                    if (ones < 64) { u += ones; br.skip(ones + 1); break; }
                    u += 64;
                    br.skip(64);
                    if (br.over()) { status = KS_INSUFFICIENT; break; }
                }
                if (status != KS_OK) break;
                u &= 255;
                const u8 v = tmtf[u];
                for (int j = u; j > 0; j--) tmtf[j] = tmtf[j - 1];
                tmtf[0] = v;
                sel[i] = v;
            }
            if (status != KS_OK) break;
            if (br.over()) { status = KS_INSUFFICIENT; break; }
            // delta-coded code lengths, :246-254
            for (int t = 0; t < ntab && status == KS_OK; t++) {
                int cur = (int)br.bits(5);
                for (int i = 0; i <= eob; i++) {
                    for (;;) {
                        br.fill();
                        if (!br.peek(1)) { br.skip(1); break; }
                        cur += (br.peek(2) & 1u) ? -1 : 1;
                        br.skip(2);
                        if (br.over()) break;
                    }
                    if (br.over()) { status = KS_INSUFFICIENT; break; }
                    if (cur < 0 || cur > KD_MAXLEN) { status = KS_TABLES; break; }
                    lens[t][i] = (u8)cur;
                }
            }
            if (status != KS_OK) break;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            // canonical decode tables, S/BZip2HuffmanStageDecoder.swift:84-141
            const int alpha = eob + 1;
            for (int t = 0; t < ntab; t++) {
                int mn = KD_MAXLEN, mx = 0;
                for (int i = 0; i < alpha; i++) {
                    const int l = lens[t][i];
                    mx = max(mx, l);
                    mn = min(mn, l);
                }
                minlen[t] = mn;
                for (int i = 0; i < alpha; i++) bases[t][lens[t][i] + 1]++;
                for (int i = 1; i < KD_MAXLEN + 2; i++) bases[t][i] += bases[t][i - 1];
                int code = 0;
                for (int i = mn; i <= mx; i++) {
                    const int base = code;
                    code += bases[t][i + 1] - bases[t][i];
                    bases[t][i] = base - bases[t][i];
                    limits[t][i] = code - 1;
                    code <<= 1;
                }
                int ci = 0;
                for (int bl = mn; bl <= mx; bl++)
                    for (int sy = 0; sy < alpha; sy++)
                        if (lens[t][sy] == bl) symbols[t][ci++] = (u16)sy;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            // the reference's search (S/BZip2HuffmanStageDecoder.swift:164-176), run once for every
            // 10-bit pattern: what it finds within 10 bits goes into a table, the rest walks the limits
            for (int t = 0; t < 6; t++) {
                for (int x = lane; x < (1 << KD_LUT_BITS); x += 64) {
                    u32 e = 0;
                    for (int cl = minlen[t]; cl <= KD_LUT_BITS; cl++) {
                        if (cl == 0) break; // a table the block does not define: walk
                        const int code = x >> (KD_LUT_BITS - cl);
                        if (code <= limits[t][cl]) {
                            const int ix = code - bases[t][cl];
                            const u32 sy = (ix >= 0 && ix < MAX_ALPHA) ? symbols[t][ix] : 511u; // 511: no symbol
                            e = (sy << 5) | (u32)cl;
                            break;
                        }
                    }
                    lut[t][x] = (u16)e;
                }
            }
        } while (0);
        if (lane == 0) { s_status = status; s_eob = (u32)eob; s_nsel = (u32)nsel; s_orig = orig; s_crc = crc; }
        // the initial list (the block's bytes in order) for kd_imtf_compose
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        ((u32 *)(symmap_out + (size_t)b * 256))[lane] = ((const u32 *)symmap)[lane];
    }
    __syncthreads();
    const u32 status0 = s_status;
    // The two serial loops below set the decoder's time, and waves of the later stages of other blocks
    // (the other stream) now share their SIMDs: highest wave priority (measured: the long group's
    // Huffman stage 85.1 -> 82.1 ms, what it takes with the chip to itself).
    __builtin_amdgcn_s_setprio(3);
    if (status0 == KS_OK) {
        const int eob = __builtin_amdgcn_readfirstlane((int)s_eob);
        if (wv == 0) {
            // ---- producer: Huffman symbols, S/BZip2HuffmanStageDecoder.swift:149-181.
            // Every lane looks up the code that would start at bit offset `lane` of the buffer (one
            // LDS access for the whole window); the true chain of codes is then followed through
            // those entries with scalar reads, several symbols per window, and the symbols of a
            // window go to the ring in one store.
            const int nsel = __builtin_amdgcn_readfirstlane((int)s_nsel);
            u32 head = 0, tail_seen = 0, published = 0;
            const u64 nbits = nbytes * 8u;
            int gleft = 0, gidx = -1, cur = sel[0];
            u32 stop = 0; // 0: running, 1: the final ring entry is among those being written, 2: gave up
            // From here on the reader is one number: bp, the next symbol's bit offset inside the LDS
            // window (which starts at stream word wbase).  Every lane fetches the 32 bits at offset
            // bp + lane straight from the window -- no scalar bit buffer to shift and refill, which
            // was a third of the loop's instructions.
            u64 wbase;
            u32 bp;
            {
                const u64 cons = br.consumed();
                wbase = KdBits::uni64(cons >> 5);
                bp = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(cons & 31u));
                br.load_window(wbase);
            }
            // a step is "near the end" when fewer than 256 bits of the stream lie beyond bp: as a
            // window-relative bound, recomputed with the window
            auto near_bound = [&](u64 wb) -> u32 {
                const u64 w0 = wb * 32u + 256u;
                return nbits > w0 ? (u32)min(nbits - w0, (u64)0xfffffffeu) + 1u : 0u;
            };
            u32 nearbp = near_bound(wbase); // bp >= nearbp: near the end
            while (!stop) {
                // wave-uniform state in scalar registers (see KdBits::pin)
                wbase = KdBits::uni64(wbase);
                bp = (u32)__builtin_amdgcn_readfirstlane((int)bp);
                nearbp = (u32)__builtin_amdgcn_readfirstlane((int)nearbp);
                gleft = __builtin_amdgcn_readfirstlane(gleft);
                gidx = __builtin_amdgcn_readfirstlane(gidx);
                cur = __builtin_amdgcn_readfirstlane(cur);
                head = (u32)__builtin_amdgcn_readfirstlane((int)head);
                tail_seen = (u32)__builtin_amdgcn_readfirstlane((int)tail_seen);
                published = (u32)__builtin_amdgcn_readfirstlane((int)published);
                u32 mysym = 0, mysym1 = 0, cnt = 0;
                u64 visited = 0, visited1 = 0;
                bool fast = false;
                if (gleft == 0) {
                    gleft = GROUP;
                    gidx++;
                    if (gidx == nsel) { mysym = KD_SYM_ERR | KS_DECODE; cnt = 1; stop = 1; }
                    else {
                        cur = __builtin_amdgcn_readfirstlane((int)sel[gidx]);
                        if (cur >= 6) { mysym = KD_SYM_ERR | KS_DECODE; cnt = 1; stop = 1; }
                    }
                }
                if (!stop) {
                    // the lanes read the bits [bp, bp + 127 + 32) and one word more
                    if (bp + 256u > KD_WIN * 8u) {
                        wbase += bp >> 5;
                        bp &= 31u;
                        br.load_window(wbase);
                        nearbp = near_bound(wbase);
                    }
                    const u32 a = bp + (u32)lane;
                    const u32 d0 = __builtin_bswap32(win[a >> 5]), d1 = __builtin_bswap32(win[(a >> 5) + 1]);
                    const u32 fld = (u32)((((u64)d0 << 32) | (u64)d1) >> (32u - (a & 31u))); // 32 bits at offset a
                    const u32 myent = lut[cur][fld >> (32 - KD_LUT_BITS)];
                    // (no bit past the end can be used by this step while 256 more bits exist)
                    const bool nearend = bp >= nearbp;
                    u32 off = 0;
                    if (!nearend) {
                        // Two registers of entries per step, for the 128 bit offsets from bp on (the
                        // step's fixed cost is as much as the chain through ~8 symbols).
                        // Entries the scalar chain stops at: no entry, no symbol, end of block; the
                        // chain follows the next offset stored in the entry (bits 16..22) and leaves a
                        // register after the entry whose successor lies outside its 64 offsets (bit 23)
                        const u32 d2 = __builtin_bswap32(win[(a >> 5) + 2]), d3 = __builtin_bswap32(win[(a >> 5) + 3]);
                        const u32 fld1 = (u32)((((u64)d2 << 32) | (u64)d3) >> (32u - (a & 31u))); // offset a + 64
                        const u32 myent1 = lut[cur][fld1 >> (32 - KD_LUT_BITS)];
                        const u32 sy = myent >> 5, sy1 = myent1 >> 5;
                        const u32 nx = (u32)lane + (myent & 31u), nx1 = (u32)lane + (myent1 & 31u);
                        const u32 ment = (myent & 0xffffu) | (nx << 16) | (nx > 63u ? 0x800000u : 0u) |
                                         ((myent == 0 || sy == 511u || (int)sy == eob) ? 0x8000u : 0u);
                        const u32 ment1 = (myent1 & 0xffffu) | (nx1 << 16) | (nx1 > 63u ? 0x800000u : 0u) |
                                          ((myent1 == 0 || sy1 == 511u || (int)sy1 == eob) ? 0x8000u : 0u);
                        u32 n = 0;
                        const u32 glim = (u32)gleft;
                        const u32 e = kd_chain2(ment, off, n, glim, visited);
                        if (n > 0) {
                            // the lanes whose offset the chain visited hold its symbols, in order
                            fast = true;
                            mysym = sy;
                            if (!(e & 0x8000u) && (e & 0x800000u) && n < glim) {
                                // on into the second register: offsets 64 ..
                                u32 off1 = off - 64u, n1 = 0;
                                (void)kd_chain2(ment1, off1, n1, glim - n, visited1);
                                off = 64u + off1;
                                n += n1;
                                mysym1 = sy1;
                            }
                            gleft -= (int)n;
                            cnt = n;
                        } else {
                            const u32 e0 = __builtin_amdgcn_readlane(myent, 0);
                            if (e0 != 0) { // no symbol for this code, or the end of the block
                                off = e0 & 31u;
                                gleft--;
                                mysym = ((e0 >> 5) == 511u) ? (KD_SYM_ERR | KS_DECODE) : (e0 >> 5);
                                cnt = 1;
                                stop = 1;
                            }
                        }
                    } else {
                        // the last bits of the stream: test every symbol for the end of the data
                        const u64 here = wbase * 32u + (u64)bp; // stream bit of the next symbol
                        while (gleft > 0 && off <= 63u) {
                            const u32 e = __builtin_amdgcn_readlane(myent, (int)off);
                            if (e == 0) break; // longer than 10 bits (or a table the block does not define)
                            off += e & 31u;
                            gleft--;
                            u32 entry = e >> 5;
                            if (here + off > nbits) { entry = KD_SYM_ERR | KS_INSUFFICIENT; stop = 1; }
                            else if (entry == 511u) { entry = KD_SYM_ERR | KS_DECODE; stop = 1; }
                            else if ((int)entry == eob) stop = 1;
                            if ((u32)lane == cnt) mysym = entry;
                            cnt++;
                            if (stop) break;
                        }
                    }
                    bp += off;
                    if (cnt == 0) {
                        // walk the limits for this one symbol: the 32 bits at bp are lane 0's field
                        // (nothing was consumed in this step)
                        const u32 f0 = __builtin_amdgcn_readlane(fld, 0);
                        gleft--;
                        int sym = -1;
                        int cl = __builtin_amdgcn_readfirstlane(minlen[cur]);
                        while (cl <= KD_MAXLEN) {
                            const int code = cl ? (int)(f0 >> (32 - cl)) : 0;
                            if (code <= __builtin_amdgcn_readfirstlane(limits[cur][cl])) {
                                const int ix = code - __builtin_amdgcn_readfirstlane(bases[cur][cl]);
                                if (ix >= 0 && ix < MAX_ALPHA) sym = __builtin_amdgcn_readfirstlane((int)symbols[cur][ix]);
                                break;
                            }
                            cl++;
                        }
                        bp += (u32)min(cl, KD_MAXLEN + 1);
                        if (nearend && wbase * 32u + (u64)bp > nbits) { mysym = KD_SYM_ERR | KS_INSUFFICIENT; stop = 1; }
                        else if (sym < 0) { mysym = KD_SYM_ERR | KS_DECODE; stop = 1; }
                        else {
                            mysym = (u32)sym;
                            if (sym == eob) stop = 1;
                        }
                        cnt = 1;
                    }
                }
                if (head + cnt - tail_seen > KD_RING) {
                    // ring full: wait for the consumer (or for its abort)
                    u32 spins = 0;
                    for (;;) {
                        tail_seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&s_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                        KD_LDS_ORDER();
                        if (head + cnt - tail_seen <= KD_RING) break;
                        if (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&s_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) ||
                            ++spins > KD_SPIN_MAX) { stop = 2; break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    if (stop == 2) break;
                }
                if (fast) {
                    const u64 ltm = (1ull << lane) - 1ull;
                    if ((visited >> lane) & 1ull)
                        ring[(head + (u32)__popcll(visited & ltm)) % KD_RING] = (u16)mysym;
                    if ((visited1 >> lane) & 1ull)
                        ring[(head + (u32)__popcll(visited) + (u32)__popcll(visited1 & ltm)) % KD_RING] = (u16)mysym1;
                } else if ((u32)lane < cnt) {
                    ring[(head + lane) % KD_RING] = (u16)mysym;
                }
                head += cnt;
                if (head - published >= 32u || stop) {
                    published = head;
                    KD_LDS_ORDER();
                    __hip_atomic_store(&s_head, head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            if (lane == 0) {
                const u64 e = wbase * 32u + (u64)bp;
                s_end_lo = (u32)e;
                s_end_hi = (u32)(e >> 32);
            }
        } else {
            // ---- consumer: symbols -> BWT bytes, S/BZip2BlockDecompressor.swift:266-321.  The MTF
            // list holds the bytes themselves (the symbol map applied to the identity list).
            // A batch of up to 64 ring entries is handled in the vector domain (lane = entry): what
            // every entry emits (a plain symbol 1
            // byte, the j-th symbol of a run stretch (1 + RUNB) << j copies of the current byte -- all
            // bytes of a stretch are equal, so each run symbol writes its own share), the output
            // offsets (a wave scan), the reference's size checks (they are monotone in those prefix
            // sums: the first lane that violates one is where the reference stops) and the stores.
            // Nothing here is serial over the symbols: a lone wave issues one instruction every
            // 6.5-10 cycles, so the serial list moves are left to kd_imtf_*.
            u32 len = 0;
            u32 runk = 0, runrep = 0; // an open run stretch: symbols so far, length so far
            u32 tail = 0, status = KS_OK;
            bool done = false;
            const u64 lt = (1ull << lane) - 1ull;
            while (!done) {
                // wait for symbols
                u32 head, spins = 0;
                for (;;) {
                    head = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&s_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    KD_LDS_ORDER();
                    if (head != tail) break;
                    if (++spins > KD_SPIN_MAX) { status = KS_DECODE; done = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (done) break;
                const u32 avail = min(head - tail, 64u);
                const bool valid = (u32)lane < avail;
                const u32 mine = valid ? ring[(tail + lane) % KD_RING] : 0u;
                const bool iserr = valid && (mine & KD_SYM_ERR);
                const bool isrun = valid && mine < 2u;
                const bool iseob = valid && !iserr && (int)mine == eob;
                const bool isplain = valid && !isrun && !iserr && !iseob;
                const u64 runm = __ballot(isrun);
                const u64 nonrun = __ballot(valid && !isrun);
                // run symbols: exponent = index inside the stretch (a stretch may come over from the
                // batch before), share = (1 + RUNB) << exponent, saturated (2^22 is already too long)
                const u64 nrb = nonrun & lt;
                const int lastnr = nrb ? 63 - __clzll((unsigned long long)nrb) : -1;
                const u32 expo = (u32)(lane - lastnr - 1) + (lastnr < 0 ? runk : 0u);
                const u32 share = isrun ? (expo < 21u ? (1u + mine) << expo : (1u << 22)) : 0u;
                const u32 e = isrun ? share : (isplain ? 1u : 0u);
                const u32 inc = wave_incl_scan_add(e);
                const u32 E = inc - e; // bytes the batch emits before this entry
                // length of the stretch up to and including this run symbol (S/BZip2BlockDecompressor.swift:
                // 286-297 checks it symbol by symbol)
                const u32 Ea = lastnr < 0 ? 0u : (u32)__shfl((int)inc, lastnr < 0 ? 0 : lastnr, 64);
                const u32 R = inc - Ea + (lastnr < 0 ? runrep : 0u);
                // the reference's stops, in its order at one symbol: error entry; run too long; a run
                // that ends here does not fit; end of block; no room for this symbol
                const bool afterrun = lane == 0 ? runk > 0 : ((runm >> (lane - 1)) & 1ull);
                u32 st = KS_OK;
                if (iserr) st = mine & 0xffu;
                else if (isrun) { if (R > 2u * 900000u + 2u) st = KS_BLOCK_SIZE; }
                else if (valid) {
                    if (afterrun && len + E > block_size) st = KS_BLOCK_SIZE;
                    else if (isplain && len + E >= block_size) st = KS_BLOCK_SIZE;
                }
                const u64 stopm = __ballot(st != KS_OK || iseob);
                const u32 fstop = stopm ? (u32)__builtin_ctzll(stopm) : 64u; // entries before it are processed
                const u64 live = fstop >= 64u ? ~0ull : ((1ull << fstop) - 1ull);
                // ---- what is written is the MTF position of every output byte: a plain symbol's
                // own (>= 1), 0 for the bytes of a run (a run repeats the front of the list, and RUNA /
                // RUNB stand for runs of position 0).  The list itself is undone afterwards, for all
                // blocks and all their chunks in parallel (kd_imtf_*).
                const u32 byte = isplain ? ((mine - 1u) & 255u) : 0u;
                const bool act = (live >> lane) & 1ull;
                const u32 o = len + E;
                const u32 cnt = act ? min(e, o < block_size ? block_size - o : 0u) : 0u; // never past the slot
                // short shares by their own lane, long ones (> 64 bytes) by the whole wave
                const u32 small = cnt <= 64u ? cnt : 0u;
                const u32 maxs = wave_reduce_max(small);
                for (u32 r = 0; r < maxs; r++)
                    if (r < small) out[o + r] = (u8)byte;
                u64 big = __ballot(cnt > 64u);
                while (big) {
                    const int q = __builtin_ctzll(big);
                    big &= big - 1ull;
                    const u32 bo = __builtin_amdgcn_readlane(o, q), bc = __builtin_amdgcn_readlane(cnt, q);
                    const u32 bv = __builtin_amdgcn_readlane(byte, q);
                    for (u32 r = (u32)lane; r < bc; r += 64u) out[bo + r] = (u8)bv;
                }
                // ---- state for the next batch
                const u32 nproc = min(fstop, avail);
                if (nproc > 0) {
                    len += __builtin_amdgcn_readlane(inc, (int)nproc - 1);
                    if ((runm >> (nproc - 1)) & 1ull) {
                        runk = __builtin_amdgcn_readlane(expo, (int)nproc - 1) + 1u;
                        runrep = __builtin_amdgcn_readlane(R, (int)nproc - 1);
                    } else {
                        runk = 0;
                        runrep = 0;
                    }
                }
                if (fstop < avail) {
                    // stopped inside the batch: an error, or the end of the block
                    status = __builtin_amdgcn_readlane(st, (int)fstop);
                    done = true;
                }
                tail += avail;
                KD_LDS_ORDER();
                __hip_atomic_store(&s_tail, tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (lane == 0) {
                s_len = len;
                s_status = status;
                __hip_atomic_store(&s_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        mt[KM_STATUS] = s_status;
        mt[KM_N] = s_len;
        mt[KM_ORIG] = s_orig;
        mt[KM_CRC] = s_crc;
        mt[KM_END_LO] = s_end_lo;
        mt[KM_END_HI] = s_end_hi;
    }
}

// ------------------------------------------------------------------ inverse MTF, parallel
// kd_huff leaves the MTF position of every output byte.  Undoing the list is serial along a
// block, but what a stretch of positions does to the list does not depend on the list's contents:
// it is a permutation of the 256 places.  So a block is cut into chunks of KD_MCHUNK positions and
//   kd_imtf_perm    : every chunk applies its moves to the identity -> its permutation
//   kd_imtf_compose : one wave per block walks the chunks: entry list of chunk c + 1 =
//                     entry list of chunk c, permuted (a 256-byte gather), from the block's bytes
//   kd_imtf_apply   : every chunk replays its moves on its entry list and writes the bytes
// (the mirror image of the encoder's k3_mtf_recency / compose / replay).  The list is 4 entries
// per lane, entry 0 = front, as bytes of a word.
#define KD_MCHUNK 1024

// The moves of one row of 64 positions (lane = position, 0 = none).  WANT: returns the lane's byte.
template <bool WANT>
__device__ __forceinline__ u32 kd_imtf_row(u32 &L, u32 p, bool valid) {
    const int lane = lane_id();
    const u32 front0 = (u32)__builtin_amdgcn_readlane((int)L, 0) & 255u; // a position 0 repeats the front
    const u32 pk = (p >> 2) | ((p & 3u) << 11); // lane | byte shift << 8
    const u64 nzm = __ballot(valid && p != 0);
    u64 nz = nzm;
    u32 vals = 0;
    while (nz) {
        const int q = __builtin_ctzll(nz);
        asm("s_bitset0_b64 %0, %1" : "+s"(nz) : "s"(q));
        const u32 k2 = __builtin_amdgcn_readlane(pk, q);
        const int src = (int)(k2 & 255u);
        const u32 sh = k2 >> 8;
        const u32 m = 0xffffffffu >> (24u - sh); // bytes 0..k of the source lane's word move
        u32 v;
        if (src == 0) {
            // one of the first four entries: lane 0's word in scalar arithmetic
            u32 F = (u32)__builtin_amdgcn_readlane((int)L, 0);
            v = (F >> sh) & 255u;
            F = (((F << 8) | v) & m) | (F & ~m);
            L = write_lane(L, F, 0);
        } else {
            v = (__builtin_amdgcn_readlane(L, src) >> sh) & 255u;
            // every word takes the top byte of the word before it, the first one v (wave_shr leaves
            // lane 0 with the `old` operand)
            const u32 up = (u32)__builtin_amdgcn_update_dpp((int)(v << 24), (int)L, 0x138, 0xf, 0xf, false);
            const u32 shifted = __builtin_amdgcn_alignbit(L, up, 24); // (L << 8) | (up >> 24)
            const u32 at_src = (shifted & m) | (L & ~m);
            L = (lane < src) ? shifted : ((lane == src) ? at_src : L);
        }
        if (WANT) vals = write_lane(vals, v, q);
    }
    if (!WANT) return 0;
    // the byte of a position 0 is that of the nearest move at or before it (or the row's first front)
    const u64 upto = nzm & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    const int s = upto ? 63 - __clzll((unsigned long long)upto) : 0;
    const u32 pv = (u32)__shfl((int)vals, s, 64);
    return upto ? pv : front0;
}

__global__ __launch_bounds__(256) void kd_imtf_perm(const u8 *__restrict__ bwt, const u32 *__restrict__ slot,
                                                    const u32 *__restrict__ nArr, u32 S, u32 CH,
                                                    u8 *__restrict__ perm) {
    const u32 k = slot[blockIdx.y], n = nArr[k];
    const int lane = lane_id();
    const u32 c = blockIdx.x * 4u + (u32)wave_id();
    const u32 base = c * KD_MCHUNK;
    if (base >= n) return;
    const u8 *pos = bwt + (size_t)k * S;
    u8 rowv[KD_MCHUNK / 64];
#pragma unroll
    for (int r = 0; r < KD_MCHUNK / 64; r++) rowv[r] = pos[min(base + r * 64 + lane, n - 1)];
    const u32 l4 = (u32)lane * 4u;
    u32 L = l4 | ((l4 + 1u) << 8) | ((l4 + 2u) << 16) | ((l4 + 3u) << 24);
#pragma unroll
    for (int r = 0; r < KD_MCHUNK / 64; r++) {
        const u32 j = base + r * 64 + lane;
        if (base + r * 64 >= n) break;
        (void)kd_imtf_row<false>(L, rowv[r], j < n);
    }
    ((u32 *)(perm + ((size_t)k * CH + c) * 256))[lane] = L;
}

// one wave per block: perm[c] := entry list of chunk c
__global__ __launch_bounds__(64) void kd_imtf_compose(const u32 *__restrict__ slot, const u32 *__restrict__ nArr,
                                                      const u8 *__restrict__ symmap, u32 CH,
                                                      u8 *__restrict__ perm) {
    const u32 k = slot[blockIdx.x], n = nArr[k];
    const int lane = lane_id();
    const u32 chunks = (n + KD_MCHUNK - 1) / KD_MCHUNK;
    __shared__ __attribute__((aligned(16))) u8 cur[256];
    ((u32 *)cur)[lane] = ((const u32 *)(symmap + (size_t)k * 256))[lane];
    __syncthreads();
    u32 *pw = (u32 *)(perm + (size_t)k * CH * 256);
    // the permutations do not depend on the walk: fetch them 8 chunks at a time
    constexpr int PF = 8;
    for (u32 c0 = 0; c0 < chunks; c0 += PF) {
        u32 pv[PF];
#pragma unroll
        for (int q = 0; q < PF; q++) pv[q] = (c0 + q < chunks) ? pw[(size_t)(c0 + q) * 64 + lane] : 0u;
#pragma unroll
        for (int q = 0; q < PF; q++) {
            if (c0 + q >= chunks) break;
            const u32 w = pv[q];
            const u32 now = ((u32 *)cur)[lane];
            pw[(size_t)(c0 + q) * 64 + lane] = now; // the chunk's entry list
            const u32 nv = (u32)cur[w & 255u] | ((u32)cur[(w >> 8) & 255u] << 8) |
                           ((u32)cur[(w >> 16) & 255u] << 16) | ((u32)cur[w >> 24] << 24);
            __syncthreads();
            ((u32 *)cur)[lane] = nv;
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void kd_imtf_apply(u8 *__restrict__ bwt, const u32 *__restrict__ slot,
                                                     const u32 *__restrict__ nArr, u32 S, u32 CH,
                                                     const u8 *__restrict__ perm) {
    const u32 k = slot[blockIdx.y], n = nArr[k];
    const int lane = lane_id();
    const u32 c = blockIdx.x * 4u + (u32)wave_id();
    const u32 base = c * KD_MCHUNK;
    if (base >= n) return;
    u8 *pos = bwt + (size_t)k * S;
    u8 rowv[KD_MCHUNK / 64];
#pragma unroll
    for (int r = 0; r < KD_MCHUNK / 64; r++) rowv[r] = pos[min(base + r * 64 + lane, n - 1)];
    u32 L = ((const u32 *)(perm + ((size_t)k * CH + c) * 256))[lane];
#pragma unroll
    for (int r = 0; r < KD_MCHUNK / 64; r++) {
        const u32 j = base + r * 64 + lane;
        if (base + r * 64 >= n) break;
        const u32 v = kd_imtf_row<true>(L, rowv[r], j < n);
        if (j < n) pos[j] = (u8)v;
    }
}

// ------------------------------------------------------------------ inverse BWT
// Every per-block array of the stages below is indexed by the candidate's slot in the batch; a launch
// works on a list of slots (`slot` / `list`), nArr / origArr come from kd_set_meta
#define KD_MPARTS 8 // the counting sort of a block runs as 8 waves over 8 index ranges
__device__ __forceinline__ u32 kd_part_len(u32 n) { return (((n + KD_MPARTS - 1) / KD_MPARTS) + 255u) & ~255u; }

// cnt[part][v] = number of bytes v among the indices of that part (one workgroup per part)
__global__ __launch_bounds__(256) void kd_hist(const u8 *__restrict__ bwt, const u32 *__restrict__ slot,
                                               const u32 *__restrict__ nArr, u32 S,
                                               u32 *__restrict__ cntArr) {
    const u32 b = slot[blockIdx.y], part = blockIdx.x;
    // BWT bytes come in runs, and lanes of one LDS atomic that meet on a word are served one by one
    // (94 % of this kernel's LDS cycles): 16 consecutive bytes per thread, equal neighbours added at
    // once, every counter 8 times (lane mod 8)
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    __shared__ u32 h[256 * 8];
    for (u32 q = threadIdx.x; q < 256 * 8; q += 256) h[q] = 0;
    __syncthreads();
    const u8 *p = bwt + (size_t)b * S;
    const u32 nn = nArr[b], pl = kd_part_len(nn);
    const u32 n = min(nn, (part + 1u) * pl);
    u32 *const mh = h + (threadIdx.x & 7u);
    // part starts are multiples of 256 and S of 4096: 16-byte loads stay aligned and inside the slot
    for (u32 i = part * pl + threadIdx.x * 16u; i < n; i += 256u * 16u) {
        const u32x4 v = *(const u32x4 *)(p + i);
        const u32 wv[4] = {v.x, v.y, v.z, v.w};
        u32 prev = wv[0] & 255u, run = 1;
#pragma unroll
        for (int q = 1; q < 16; q++) {
            const u32 d = (wv[q >> 2] >> (8 * (q & 3))) & 255u;
            const bool in = i + q < n;
            if (in && d != prev) {
                atomicAdd(&mh[prev * 8u], run);
                prev = d;
                run = 0;
            }
            run += in ? 1u : 0u;
        }
        atomicAdd(&mh[prev * 8u], run);
    }
    __syncthreads();
    u32 sum = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) sum += h[threadIdx.x * 8 + r];
    cntArr[((size_t)b * KD_MPARTS + part) * 256 + threadIdx.x] = sum;
}

// merged[base[v]++] = (i << 8) + v in index order (S/BZip2BlockDecompressor.swift:346-350); one wave
// per part of the block
__global__ __launch_bounds__(64) void kd_merge(const u8 *__restrict__ bwt, const u32 *__restrict__ slot,
                                               const u32 *__restrict__ nArr, u32 S,
                                               const u32 *__restrict__ base, u32 *__restrict__ merged) {
    const u32 b = slot[blockIdx.y], part = blockIdx.x;
    const int lane = lane_id();
    __shared__ u32 run[256];
    {
        // first merged slot of byte v for this part = bytes below v in the whole block + bytes v in
        // the parts before this one (lane holds v = 4 lane .. 4 lane + 3)
        const uint4 *cp = (const uint4 *)(base + (size_t)b * KD_MPARTS * 256) + lane;
        u32 tot[4] = {0, 0, 0, 0}, before[4] = {0, 0, 0, 0};
#pragma unroll
        for (u32 q = 0; q < KD_MPARTS; q++) {
            const uint4 c4 = cp[q * 64];
            const u32 cc[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                tot[j] += cc[j];
                if (q < part) before[j] += cc[j];
            }
        }
        const u32 mine = tot[0] + tot[1] + tot[2] + tot[3];
        u32 ex = wave_incl_scan_add(mine) - mine;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            run[lane * 4 + j] = ex + before[j];
            ex += tot[j];
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const u8 *p = bwt + (size_t)b * S;
    u32 *mg = merged + (size_t)b * S;
    const u32 nn = nArr[b], pl = kd_part_len(nn);
    const u32 n = min(nn, (part + 1u) * pl); // this part: [part * pl, n)
    const u64 lt = (1ull << lane) - 1ull;
    for (u32 r0 = part * pl; r0 < n; r0 += 256) {
        u32 vv[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 i = r0 + k * 64 + lane;
            vv[k] = (i < n) ? p[i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 i = r0 + k * 64 + lane;
            const bool act = i < n;
            const u32 v = vv[k];
            u64 m = __ballot(act);
#pragma unroll
            for (int bit = 0; bit < 8; bit++) {
                const bool s = (v >> bit) & 1u;
                const u64 bm = __ballot(s);
                m &= s ? bm : ~bm;
            }
            if (act) {
                const u32 rk = (u32)__popcll(m & lt);
                const u32 bs = run[v];
                mg[bs + rk] = (i << 8) + v;
                if (rk == 0) run[v] = bs + (u32)__popcll(m);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

__device__ __forceinline__ bool kd_is_ruler(u32 x, u32 orig) { return (x % KD_RULER) == 0 || x == orig; }
__device__ __forceinline__ u32 kd_ruler_id(u32 x, u32 orig, u32 NR) {
    return (x % KD_RULER) == 0 ? x / KD_RULER : NR; // the origin, when it is not a regular ruler
}

// a) every ruler walks to the next ruler: rinfo[id] = {next ruler id, steps}
__global__ __launch_bounds__(256) void kd_walk_a(const u32 *__restrict__ merged,
                                                 const u32 *__restrict__ nArr,
                                                 const u32 *__restrict__ origArr, u32 S, u32 RS,
                                                 uint2 *__restrict__ rinfo, const u32 *__restrict__ list) {
    const u32 b = list[blockIdx.y];
    const u32 n = nArr[b], orig = origArr[b];
    if (n == 0) return;
    const u32 NR = (n + KD_RULER - 1) / KD_RULER;
    const u32 id = blockIdx.x * 256 + threadIdx.x;
    if (id > NR) return;
    u32 x;
    if (id < NR) x = id * KD_RULER;
    else {
        if (orig % KD_RULER == 0) return;
        x = orig;
    }
    const u32 *mg = merged + (size_t)b * S;
    u32 steps = 0;
    do {
        x = mg[x] >> 8;
        steps++;
    } while (!kd_is_ruler(x, orig) && steps < n);
    rinfo[(size_t)b * RS + id] = make_uint2(kd_ruler_id(x, orig, NR), steps);
}

// b) one lane per block: the ruler chain from the origin; visits[k] = {ruler id, output offset}
__global__ void kd_walk_b(const u32 *__restrict__ nArr, const u32 *__restrict__ origArr, u32 RS,
                          const uint2 *__restrict__ rinfo, uint2 *__restrict__ visits,
                          u32 *__restrict__ nvis, u32 *__restrict__ cyc, u32 nb,
                          const u32 *__restrict__ list) {
    const u32 bi = blockIdx.x * blockDim.x + threadIdx.x;
    if (bi >= nb) return;
    const u32 b = list[bi];
    const u32 n = nArr[b], orig = origArr[b];
    if (n == 0) { nvis[b] = 0; cyc[b] = 0; return; }
    const u32 NR = (n + KD_RULER - 1) / KD_RULER;
    const u32 start = kd_ruler_id(orig, orig, NR);
    const uint2 *ri = rinfo + (size_t)b * RS;
    uint2 *vs = visits + (size_t)b * RS;
    u32 off = 0, r = start, v = 0;
    do {
        vs[v++] = make_uint2(r, off);
        const uint2 e = ri[r];
        off += e.y;
        r = e.x;
    } while (off < n && r != start && v < RS);
    nvis[b] = v;
    cyc[b] = min(off, n); // < n: the walk closed early (block = w^k); the rest repeats
}

// c) every visited ruler walks again and writes its bytes
__global__ __launch_bounds__(256) void kd_walk_c(const u32 *__restrict__ merged,
                                                 const u32 *__restrict__ nArr,
                                                 const u32 *__restrict__ origArr, u32 S, u32 RS,
                                                 const uint2 *__restrict__ rinfo,
                                                 const uint2 *__restrict__ visits,
                                                 const u32 *__restrict__ nvis,
                                                 const u32 *__restrict__ cyc, u8 *__restrict__ tt,
                                                 const u32 *__restrict__ list) {
    const u32 b = list[blockIdx.y];
    const u32 k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nvis[b]) return;
    const u32 n = nArr[b], orig = origArr[b];
    const u32 NR = (n + KD_RULER - 1) / KD_RULER;
    const uint2 vi = visits[(size_t)b * RS + k];
    const u32 steps = rinfo[(size_t)b * RS + vi.x].y;
    u32 x = (vi.x < NR) ? vi.x * KD_RULER : orig;
    const u32 *mg = merged + (size_t)b * S;
    u8 *o = tt + (size_t)b * S;
    const u32 lim = cyc[b];
    for (u32 t = 0; t < steps; t++) {
        const u32 m = mg[x];
        if (vi.y + t < lim) o[vi.y + t] = (u8)(m & 0xffu);
        x = m >> 8;
    }
}

__global__ __launch_bounds__(256) void kd_cycfill(const u32 *__restrict__ nArr,
                                                  const u32 *__restrict__ cyc, u32 S,
                                                  u8 *__restrict__ tt, const u32 *__restrict__ list) {
    const u32 b = list[blockIdx.y];
    const u32 n = nArr[b], c = cyc[b];
    if (c == 0 || c >= n) return;
    u8 *o = tt + (size_t)b * S;
    for (u32 t = c + blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) o[t] = o[t % c];
}

// ------------------------------------------------------------------ final run-length decoding
// S/BZip2BlockDecompressor.swift:386-420.  After four equal bytes the next byte is a repeat count.
// Over a run of L equal bytes entered with s = "its first byte is a count", the pattern is
// [count] d d d d c d d d d c ...: byte k of the run is a count iff k >= s and (k - s) % 5 == 4, or
// k == 0 and s == 1; the next run starts with a count iff (L - s) % 5 == 4.  That is a two-state
// automaton over runs: each run end carries the function s -> s', composed by a wave scan.
__device__ __forceinline__ u32 kd_fcompose(u32 a, u32 b) { // a earlier, b later; bit s = image of s
    return ((b >> (a & 1u)) & 1u) | (((b >> ((a >> 1) & 1u)) & 1u) << 1);
}

#define KD_SEGS 64 // a block's bytes are cut at run heads into up to 64 pieces, one wave each

// One piece [a, e) of a block (a is a run head, e a run head or n), entered in state s_in with the
// output offset off: returns the state after it and advances off; WRITE stores the bytes.
template <bool WRITE>
__device__ __forceinline__ u32 kd_unrle_range(const u8 *__restrict__ p, u32 a, u32 e, u32 n, u32 S,
                                              u32 s_in, u32 &off, bool &overflow, u8 *__restrict__ o) {
    const int lane = lane_id();
    u32 carry_prev = 256, carry_rs = a;
    for (u32 rb4 = a; rb4 < e; rb4 += 256) {
      // four rows of loads in flight (clamped addresses: the slot is padded), then the rows in turn
      u32 vq[4], nq[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
          const u32 t = rb4 + r * 64 + lane;
          vq[r] = p[min(t, S - 1)];
          nq[r] = p[min(t + 1, S - 1)];
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const u32 rowbase = rb4 + r * 64;
        if (rowbase >= e) break;
        const u32 t = rowbase + lane;
        const bool act = t < e;
        const u32 v = act ? vq[r] : 0u;
        const u32 nxt = (t + 1 < n) ? nq[r] : 256u; // p[e] starts another run: the piece's last run ends at e
        u32 pv = __shfl_up(v, 1, 64);
        if (lane == 0) pv = carry_prev;
        const bool head = act && (t == a || v != pv);
        u32 hs = head ? t + 1 : 0u;
        hs = wave_incl_scan_max(hs);
        const u32 rs = max(hs, carry_rs + 1) - 1; // start of the run this byte belongs to
        const bool end = act && (nxt != v);
        const u32 Lr = t - rs + 1;
        u32 F = 2u; // identity
        if (end) F = ((Lr % 5u) == 4u ? 1u : 0u) | (((Lr - 1u) % 5u) == 4u ? 2u : 0u);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u32 g = __shfl_up(F, d, 64);
            if (lane >= d) F = kd_fcompose(g, F);
        }
        u32 Fx = __shfl_up(F, 1, 64);
        if (lane == 0) Fx = 2u;
        const u32 s_run = (Fx >> s_in) & 1u;
        const u32 k = t - rs;
        const bool is_count = act && ((s_run == 1u && k == 0u) || (k >= s_run && ((k - s_run) % 5u) == 4u));
        const u32 contrib = act ? (is_count ? v : 1u) : 0u;
        const u32 inc = wave_incl_scan_add(contrib);
        const u32 tot = __shfl(inc, 63, 64);
        if (off + tot < off) overflow = true;
        if (WRITE) {
            const u32 my = off + inc - contrib;
            if (act && !is_count) o[my] = (u8)v;
            // a count: the whole wave writes its copies (of the byte before it; for the first byte
            // of a piece that is the last byte of the piece before)
            const u32 rb = (t == a && a > 0) ? (u32)p[a - 1] : pv;
            u64 cm = __ballot(is_count && v > 0);
            while (cm) {
                const int i = __ffsll((unsigned long long)cm) - 1;
                cm &= cm - 1;
                const u32 cnt = __shfl(v, i, 64), at = __shfl(my, i, 64), byte = __shfl(rb, i, 64);
                for (u32 q = lane; q < cnt; q += 64) o[at + q] = (u8)byte;
            }
        }
        off += tot;
        s_in = (__shfl(F, 63, 64) >> s_in) & 1u;
        carry_prev = __shfl(v, 63, 64);
        carry_rs = __shfl(rs, 63, 64);
      }
    }
    return s_in;
}

// piece k of a block starts at the first run head at or after k * ceil(n / KD_SEGS)
__global__ __launch_bounds__(64) void kd_segs(const u8 *__restrict__ tt, const u32 *__restrict__ nArr,
                                              u32 S, u32 *__restrict__ segstart,
                                              const u32 *__restrict__ list) {
    const u32 b = list[blockIdx.y], k = blockIdx.x;
    const u32 n = nArr[b];
    const u8 *p = tt + (size_t)b * S;
    u32 pos = (u32)(((u64)k * n + KD_SEGS - 1) / KD_SEGS);
    u32 found = n;
    if (k == 0) found = 0;
    else {
        while (pos < n) {
            const u32 t = pos + lane_id();
            const bool hd = t < n && p[t] != p[t - 1]; // t >= 1 here
            const u64 m = __ballot(hd);
            if (m) { found = pos + (u32)__ffsll((unsigned long long)m) - 1u; break; }
            pos += 64;
        }
    }
    if (lane_id() == 0) {
        segstart[(size_t)b * (KD_SEGS + 1) + k] = found;
        if (k == 0) segstart[(size_t)b * (KD_SEGS + 1) + KD_SEGS] = n;
    }
}

// per piece and entry state: state after it and bytes it expands to
__global__ __launch_bounds__(64) void kd_unrle_count(const u8 *__restrict__ tt,
                                                     const u32 *__restrict__ nArr, u32 S,
                                                     const u32 *__restrict__ segstart,
                                                     u32 *__restrict__ segres,
                                                     const u32 *__restrict__ list) {
    const u32 b = list[blockIdx.y], k = blockIdx.x;
    const u32 n = nArr[b];
    const u8 *p = tt + (size_t)b * S;
    const u32 a = segstart[(size_t)b * (KD_SEGS + 1) + k], e = segstart[(size_t)b * (KD_SEGS + 1) + k + 1];
    u32 len0 = 0, len1 = 0;
    bool ov = false;
    const u32 x0 = kd_unrle_range<false>(p, a, e, n, S, 0u, len0, ov, nullptr);
    const u32 x1 = kd_unrle_range<false>(p, a, e, n, S, 1u, len1, ov, nullptr);
    if (lane_id() == 0) {
        u32 *r = segres + ((size_t)b * KD_SEGS + k) * 4;
        r[0] = x0 | (x1 << 1) | (ov ? 4u : 0u);
        r[1] = len0;
        r[2] = len1;
    }
}

// per block: entry state and output offset of every piece; the block's length and flags
__global__ void kd_unrle_resolve(const u32 *__restrict__ segres, u32 *__restrict__ segin,
                                 u32 *__restrict__ outlen, u32 *__restrict__ flags, u32 nb,
                                 const u32 *__restrict__ list) {
    const u32 bi = blockIdx.x * blockDim.x + threadIdx.x;
    if (bi >= nb) return;
    const u32 b = list[bi];
    u32 st = 0, off = 0;
    bool ov = false;
    for (u32 k = 0; k < KD_SEGS; k++) {
        const u32 *r = segres + ((size_t)b * KD_SEGS + k) * 4;
        segin[((size_t)b * KD_SEGS + k) * 2] = st;
        segin[((size_t)b * KD_SEGS + k) * 2 + 1] = off;
        const u32 len = st ? r[2] : r[1];
        if (off + len < off || (r[0] & 4u)) ov = true;
        off += len;
        st = (r[0] >> st) & 1u;
    }
    outlen[b] = off;
    flags[b] = (st ? 1u : 0u) | (ov ? 2u : 0u); // 1: the block ends where a count is due
}

__global__ __launch_bounds__(64) void kd_unrle_write(const u8 *__restrict__ tt,
                                                     const u32 *__restrict__ nArr, u32 S,
                                                     const u32 *__restrict__ segstart,
                                                     const u32 *__restrict__ segin,
                                                     const u32 *__restrict__ outbase,
                                                     u8 *__restrict__ out,
                                                     const u32 *__restrict__ order) {
    // blockIdx.y = position in the block chain (output order); order[] = its candidate slot
    const u32 b = order[blockIdx.y], k = blockIdx.x;
    const u32 n = nArr[b];
    const u8 *p = tt + (size_t)b * S;
    const u32 a = segstart[(size_t)b * (KD_SEGS + 1) + k], e = segstart[(size_t)b * (KD_SEGS + 1) + k + 1];
    u32 off = segin[((size_t)b * KD_SEGS + k) * 2 + 1];
    bool ov = false;
    (void)kd_unrle_range<true>(p, a, e, n, S, segin[((size_t)b * KD_SEGS + k) * 2], off, ov,
                               out + outbase[blockIdx.y]);
}

// What the stages after kd_huff need of a candidate, from its meta words: length and origin pointer
// (0 / 0 for a candidate that did not decode, or whose pointer is invalid: every stage skips it).
__global__ void kd_set_meta(const u32 *__restrict__ list, u32 count, const u32 *__restrict__ meta,
                            u32 block_size, u32 *__restrict__ nArr, u32 *__restrict__ origArr) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const u32 b = list[i];
    const u32 *mt = meta + (size_t)b * KM_WORDS;
    u32 n = mt[KM_N], orig = mt[KM_ORIG];
    if (mt[KM_STATUS] != KS_OK || n > block_size || orig >= n) { n = 0; orig = 0; }
    nArr[b] = n;
    origArr[b] = orig;
}

// The 16 stream bytes from the byte in which each decoded candidate ends (zero-filled past the end of
// the input): where a stream ends there, they hold its end marker, its CRC and the header of the stream
// that follows (at most 11 + 4 of them), which is all the host needs to walk on into that stream.
__global__ void kd_tails(const u8 *__restrict__ in, size_t nbytes, const u32 *__restrict__ meta, u32 count,
                         u8 *__restrict__ tails) {
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= count) return;
    const u32 *mt = meta + (size_t)b * KM_WORDS;
    const u64 end = (u64)mt[KM_END_LO] | ((u64)mt[KM_END_HI] << 32);
    const u64 byte0 = end >> 3;
    const bool ok = mt[KM_STATUS] == KS_OK;
    for (u32 q = 0; q < 16; q++) tails[(size_t)b * 16 + q] = (ok && byte0 + q < nbytes) ? in[byte0 + q] : (u8)0;
}

// ------------------------------------------------------------------ host driver
namespace {
int map_status(u32 s) {
    switch (s) {
    case KS_INSUFFICIENT: return BZ2HIP_E_INSUFFICIENT;
    case KS_TABLES: return BZ2HIP_E_TABLES;
    case KS_BLOCK_SIZE: return BZ2HIP_E_BLOCK_SIZE;
    case KS_DECODE: return BZ2HIP_E_DECODE;
    case KS_RANDOMISED: return BZ2HIP_E_RANDOMISED;
    default: return BZ2HIP_OK;
    }
}
// big-endian bits [bit, bit + count) of a host copy of the stream; count <= 32
bool host_bits(const std::vector<u8> &buf, u64 buf_bit0, u64 nbits_total, u64 bit, int count, u32 *out) {
    if (bit + (u64)count > nbits_total) return false;
    u64 v = 0;
    const u64 rel = bit - buf_bit0;
    for (int q = 0; q < 8; q++) {
        const u64 idx = (rel >> 3) + q;
        v = (v << 8) | (idx < buf.size() ? buf[idx] : 0);
    }
    *out = (u32)((v << (rel & 7)) >> (64 - count));
    return true;
}
} // namespace

// Decodes the stream at d_in into d_out.  Returns BZ2HIP_OK or the first error the reference would
// have thrown; *produced = bytes it would have handed out before that (written to d_out).
// Candidates are decoded KD_BATCH at a time (bounded workspace); the chain walk continues across
// batches.
#define KD_BATCH 1024
// multi: after a stream's end marker go on with the stream that follows it in d_in, as long as one of
// the same level does (concatenated streams, what pbzip2-style writers produce and the bzip2 tool
// decodes; the reference's BZip2InputStream stops at the first end marker, S/BZip2InputStream.swift:
// 201-242): the blocks of ALL streams are candidates of one scan and share the batches, so many small
// streams fill the chip like one large one.  *streams_out = streams decoded completely.
int run_decompress(bz2hip_ctx *c, const u8 *d_in, size_t nbytes, u8 *d_out, size_t cap, size_t *produced,
                   u32 *blocks_out, bool multi, u32 *streams_out) {
    *produced = 0;
    if (blocks_out) *blocks_out = 0;
    if (streams_out) *streams_out = 0;
    c->dec_consumed = 0;
    hipStream_t st = c->stream;
    const u64 total_bits = (u64)nbytes * 8u;
    // stream header, S/BZip2InputStream.swift:166-180
    u8 hdr[4] = {0, 0, 0, 0};
    if (nbytes < 4) { c->err = "Insufficient data"; return BZ2HIP_E_INSUFFICIENT; }
    HIPCHK(hipMemcpyAsync(hdr, d_in, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const int level = (int)hdr[3] - '0';
    if (hdr[0] != 'B' || hdr[1] != 'Z' || hdr[2] != 'h' || level < 1 || level > 9) {
        c->err = "Invalid BZip2 header";
        return BZ2HIP_E_HEADER;
    }
    const u32 block_size = (u32)level * 100000u;
    const u32 S = ((block_size + 64 + RTILE - 1) / RTILE) * RTILE;

    // A real block is at least ~ 130 bits long, so a well-formed stream has fewer magics than this;
    // a crafted one can repeat the 48-bit magic every 45 bits (it overlaps itself by 3).  The scan
    // counts every hit even when the list is full: on overflow the list is regrown to the count and
    // the scan repeated, so such input gets the reference's error for its first block, not NOMEM.
    u32 cand_cap = (u32)std::min<u64>(nbytes / 8 + 16, 1u << 24);
    // The workspace is one region kept in the context between calls (grown on demand): candidates
    // first (their number sets the batch size), then the batch arrays.
    const u32 RS = (block_size + KD_RULER - 1) / KD_RULER + 2;
    const u32 ETPB_MAX = (u32)(((u64)block_size * 52 + 4095) / 4096 + 2);
    u64 *d_cand = nullptr;
    u32 *d_ncand = nullptr;
    u8 *d_bwt = nullptr, *d_tt = nullptr, *d_symmap = nullptr, *d_perm = nullptr;
    const u32 CH = S / KD_MCHUNK + 1; // chunks of the inverse MTF per block slot
    u32 *d_meta = nullptr, *d_slot = nullptr, *d_n = nullptr, *d_orig = nullptr, *d_base = nullptr, *d_merged = nullptr;
    u32 *d_order = nullptr;
    u8 *d_tails = nullptr;
    uint2 *d_rinfo = nullptr, *d_visits = nullptr;
    u32 *d_nvis = nullptr, *d_cyc = nullptr, *d_outlen = nullptr, *d_flags = nullptr, *d_outbase = nullptr, *d_crc = nullptr;
    u32 *d_segstart = nullptr, *d_segres = nullptr, *d_segin = nullptr, *d_cagg = nullptr;
    auto layout = [&](u8 *base, u32 NB) -> size_t {
        Carver A{base};
        d_cand = A.get<u64>(cand_cap);
        d_ncand = A.get<u32>(4);
        d_bwt = A.get<u8>((size_t)NB * S);
        d_tt = A.get<u8>((size_t)NB * S);
        d_meta = A.get<u32>((size_t)NB * KM_WORDS);
        d_symmap = A.get<u8>((size_t)NB * 256);
        d_perm = A.get<u8>((size_t)NB * CH * 256);
        d_slot = A.get<u32>(NB); d_n = A.get<u32>(NB); d_orig = A.get<u32>(NB); d_order = A.get<u32>(NB);
        d_base = A.get<u32>((size_t)NB * KD_MPARTS * 256);
        d_merged = A.get<u32>((size_t)NB * S);
        d_rinfo = A.get<uint2>((size_t)NB * RS); d_visits = A.get<uint2>((size_t)NB * RS);
        d_nvis = A.get<u32>(NB); d_cyc = A.get<u32>(NB);
        d_outlen = A.get<u32>(NB); d_flags = A.get<u32>(NB); d_outbase = A.get<u32>(NB + 1);
        d_crc = A.get<u32>(NB);
        d_segstart = A.get<u32>((size_t)NB * (KD_SEGS + 1)); d_segres = A.get<u32>((size_t)NB * KD_SEGS * 4);
        d_segin = A.get<u32>((size_t)NB * KD_SEGS * 2);
        d_cagg = A.get<u32>((size_t)NB * ETPB_MAX);
        d_tails = A.get<u8>((size_t)NB * 16);
        return A.off;
    };
    auto ensure = [&](u32 NB) {
        const size_t need = layout(nullptr, NB);
        if (c->dec_region_cap < need) {
            c->dec_region_cap = 0;
            if (c->dec_region) (void)hipFree(c->dec_region);
            c->dec_region = nullptr;
            HIPCHK(hipMalloc(&c->dec_region, need));
            c->dec_region_cap = need;
        }
        layout((u8 *)c->dec_region, NB);
    };
    ensure(1); // the candidate list sits at the front of the region: it survives the regrowth below only
               // through the host copy, which is uploaded again once the batch size is known
    HIPCHK(hipMemsetAsync(d_ncand, 0, 16, st));
    kd_scan_magic<<<(u32)((nbytes + 255) / 256), 256, 0, st>>>(d_in, nbytes, d_cand, d_ncand, cand_cap);
    KCHECK();
    u32 ncand = 0;
    HIPCHK(hipMemcpyAsync(&ncand, d_ncand, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (ncand > cand_cap) {
        if ((u64)ncand > total_bits / 45 + 1) throw HipError(BZ2HIP_E_HIP, "internal: magic scan counted more hits than fit the input");
        cand_cap = ncand;
        ensure(1);
        HIPCHK(hipMemsetAsync(d_ncand, 0, 16, st));
        kd_scan_magic<<<(u32)((nbytes + 255) / 256), 256, 0, st>>>(d_in, nbytes, d_cand, d_ncand, cand_cap);
        KCHECK();
        HIPCHK(hipMemcpyAsync(&ncand, d_ncand, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (ncand > cand_cap) throw HipError(BZ2HIP_E_HIP, "internal: magic scan is not repeatable");
    }
    std::vector<u64> cand(ncand);
    if (ncand) {
        HIPCHK(hipMemcpyAsync(cand.data(), d_cand, ncand * sizeof(u64), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        std::sort(cand.begin(), cand.end());
    }
    // batch workspace
    const u32 NB = std::max(1u, std::min<u32>(ncand, KD_BATCH));
    ensure(NB);
    if (ncand) HIPCHK(hipMemcpyAsync(d_cand, cand.data(), ncand * sizeof(u64), hipMemcpyHostToDevice, st));
    std::vector<u32> meta((size_t)NB * KM_WORDS);

    u32 stream_crc = 0, nblocks = 0, nstreams = 0;
    size_t out_total = 0;
    u64 cur = 32;
    for (;;) { // streams (one unless `multi`)
    for (;;) {
        auto it = std::lower_bound(cand.begin(), cand.end(), cur);
        if (it == cand.end() || *it != cur) break; // not a block: end marker or garbage, below
        const u32 i0 = (u32)(it - cand.begin());
        const u32 nbatch = std::min<u32>(ncand - i0, NB);
        // The Huffman stage lasts as long as its slowest block (the chain of a block is serial), and
        // that is a block with many compressed bits.  The candidates are therefore grouped by the
        // distance to the next magic (longest first), each group on a stream of its own with
        // everything that follows per block -- inverse MTF, inverse BWT, the counting half of the
        // run-length decoding -- so the later stages of the short blocks run under the Huffman stage
        // of the long ones.  Nothing here needs the host: a candidate's length and origin pointer go
        // from its meta words to the stages on the device (a candidate off the chain is decoded for
        // nothing, as before).  Only the chain walk and the placement of the output wait for all.
        // Two groups on the context's two streams: a process has few hardware queues (4 here), and
        // two streams that share one run their groups one after the other -- with four groups the
        // fourth Huffman launch waited behind the whole chain of the third (151 ms instead of 109).
        constexpr u32 KD_GROUPS = 2;
        std::vector<u32> glist;
        glist.reserve(nbatch);
        u32 gbeg[KD_GROUPS + 1];
        {
            std::vector<u64> span(nbatch);
            u64 smax = 1;
            for (u32 ci = 0; ci < nbatch; ci++) {
                const u64 nxt = (i0 + ci + 1 < ncand) ? cand[i0 + ci + 1] : total_bits;
                span[ci] = nxt - cand[i0 + ci];
                smax = std::max(smax, span[ci]);
            }
            for (u32 g = 0; g < KD_GROUPS; g++) {
                gbeg[g] = (u32)glist.size();
                const u32 want = KD_GROUPS - 1 - g; // longest first
                for (u32 ci = 0; ci < nbatch; ci++)
                    if ((u32)std::min<u64>(KD_GROUPS - 1, span[ci] * KD_GROUPS / (smax + 1)) == want) glist.push_back(ci);
            }
            gbeg[KD_GROUPS] = (u32)glist.size();
        }
        HIPCHK(hipMemcpyAsync(d_slot, glist.data(), nbatch * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipEventRecord(c->ev_fork, st)); // also orders the candidate upload before the other streams
        hipStream_t gstream[KD_GROUPS] = {st, c->stream_other};
        for (u32 g = 0; g < KD_GROUPS; g++) {
            const u32 cnt = gbeg[g + 1] - gbeg[g];
            if (!cnt) continue;
            hipStream_t gs = gstream[g];
            if (gs != st) HIPCHK(hipStreamWaitEvent(gs, c->ev_fork, 0));
            const u32 *lst = d_slot + gbeg[g];
            kd_huff<<<cnt, 128, 0, gs>>>(d_in, nbytes, d_cand + i0, block_size, S, d_bwt, d_meta, d_symmap, lst);
            KCHECK();
            kd_set_meta<<<(cnt + 63) / 64, 64, 0, gs>>>(lst, cnt, d_meta, block_size, d_n, d_orig);
            KCHECK();
            // MTF positions -> BWT bytes, in place
            const dim3 gm((block_size + 4 * KD_MCHUNK - 1) / (4 * KD_MCHUNK), cnt);
            kd_imtf_perm<<<gm, 256, 0, gs>>>(d_bwt, lst, d_n, S, CH, d_perm);
            KCHECK();
            kd_imtf_compose<<<cnt, 64, 0, gs>>>(lst, d_n, d_symmap, CH, d_perm);
            KCHECK();
            kd_imtf_apply<<<gm, 256, 0, gs>>>(d_bwt, lst, d_n, S, CH, d_perm);
            KCHECK();
            kd_hist<<<dim3(KD_MPARTS, cnt), 256, 0, gs>>>(d_bwt, lst, d_n, S, d_base);
            KCHECK();
            kd_merge<<<dim3(KD_MPARTS, cnt), 64, 0, gs>>>(d_bwt, lst, d_n, S, d_base, d_merged);
            KCHECK();
            const dim3 gr((RS + 255) / 256, cnt);
            kd_walk_a<<<gr, 256, 0, gs>>>(d_merged, d_n, d_orig, S, RS, d_rinfo, lst);
            KCHECK();
            kd_walk_b<<<(cnt + 63) / 64, 64, 0, gs>>>(d_n, d_orig, RS, d_rinfo, d_visits, d_nvis, d_cyc, cnt, lst);
            KCHECK();
            kd_walk_c<<<gr, 256, 0, gs>>>(d_merged, d_n, d_orig, S, RS, d_rinfo, d_visits, d_nvis, d_cyc, d_tt, lst);
            KCHECK();
            kd_cycfill<<<dim3(64, cnt), 256, 0, gs>>>(d_n, d_cyc, S, d_tt, lst);
            KCHECK();
            kd_segs<<<dim3(KD_SEGS, cnt), 64, 0, gs>>>(d_tt, d_n, S, d_segstart, lst);
            KCHECK();
            kd_unrle_count<<<dim3(KD_SEGS, cnt), 64, 0, gs>>>(d_tt, d_n, S, d_segstart, d_segres, lst);
            KCHECK();
            kd_unrle_resolve<<<(cnt + 63) / 64, 64, 0, gs>>>(d_segres, d_segin, d_outlen, d_flags, cnt, lst);
            KCHECK();
        }
        for (u32 g = 0; g < KD_GROUPS; g++)
            if (gbeg[g + 1] > gbeg[g]) HIPCHK(hipStreamSynchronize(gstream[g]));
        std::vector<u32> s_outlen(nbatch), s_flags(nbatch); // per candidate slot
        std::vector<u8> s_tails;
        if (multi) {
            s_tails.resize((size_t)nbatch * 16);
            kd_tails<<<(nbatch + 63) / 64, 64, 0, st>>>(d_in, nbytes, d_meta, nbatch, d_tails);
            KCHECK();
            HIPCHK(hipMemcpyAsync(s_tails.data(), d_tails, (size_t)nbatch * 16, hipMemcpyDeviceToHost, st));
        }
        HIPCHK(hipMemcpyAsync(meta.data(), d_meta, (size_t)nbatch * KM_WORDS * sizeof(u32),
                              hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(s_outlen.data(), d_outlen, nbatch * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(s_flags.data(), d_flags, nbatch * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        // chain walk inside the batch, S/BZip2InputStream.swift:201-242
        std::vector<u32> chain; // batch slot of block k
        // streams that end inside this batch with the next one's first block in it too: block `k` of the
        // chain is the first of a new stream, `stored` the CRC the stream before it declares
        struct Bound { u32 k, stored; size_t consumed; };
        std::vector<Bound> bounds;
        int rc_huff = BZ2HIP_OK;
        for (;;) {
            auto jt = std::lower_bound(cand.begin() + i0, cand.begin() + i0 + nbatch, cur);
            if (jt == cand.begin() + i0 + nbatch || *jt != cur) {
                // A stream that ends here and is followed by one whose first block this batch decoded
                // too: walk on.  Everything else (last stream, damaged trailer, next batch) is the
                // business of the code behind the batch loop, as for a single stream.
                if (!multi || chain.empty()) break;
                const u8 *t = &s_tails[(size_t)chain.back() * 16];
                std::vector<u8> tv(t, t + 16);
                const u64 b0 = cur >> 3;
                u32 m1 = 0, m2 = 0, stored = 0;
                if (!host_bits(tv, b0 * 8, total_bits, cur, 24, &m1) || !host_bits(tv, b0 * 8, total_bits, cur + 24, 24, &m2) ||
                    m1 != 0x177245u || m2 != 0x385090u || !host_bits(tv, b0 * 8, total_bits, cur + 48, 32, &stored))
                    break;
                const size_t nb0 = (size_t)((cur + 80 + 7) / 8);
                if (nb0 + 4 > nbytes) break;
                const u8 *h = t + (nb0 - b0);
                if (h[0] != 'B' || h[1] != 'Z' || h[2] != 'h' || (int)h[3] - '0' != level) break;
                const u64 first = (u64)nb0 * 8u + 32u;
                auto nt = std::lower_bound(cand.begin() + i0, cand.begin() + i0 + nbatch, first);
                if (nt == cand.begin() + i0 + nbatch || *nt != first) break;
                bounds.push_back({(u32)chain.size(), stored, nb0});
                cur = first;
                continue;
            }
            const u32 ci = (u32)(jt - cand.begin()) - i0;
            const u32 *mt = &meta[(size_t)ci * KM_WORDS];
            if (mt[KM_STATUS] != KS_OK) { rc_huff = map_status(mt[KM_STATUS]); break; }
            chain.push_back(ci);
            cur = (u64)mt[KM_END_LO] | ((u64)mt[KM_END_HI] << 32);
        }
        const u32 nb = (u32)chain.size();
        if (nb) {
            std::vector<u32> h_crc(nb), h_outlen(nb), h_flags(nb);
            u32 first_bad_ptr = nb; // S/BZip2BlockDecompressor.swift:332-334
            for (u32 k = 0; k < nb; k++) {
                const u32 *mt = &meta[(size_t)chain[k] * KM_WORDS];
                h_crc[k] = mt[KM_CRC];
                if (mt[KM_ORIG] >= mt[KM_N] && first_bad_ptr == nb) first_bad_ptr = k;
                h_outlen[k] = s_outlen[chain[k]];
                h_flags[k] = s_flags[chain[k]];
            }
            // the chain in output order, for the stage that writes (kd_unrle_write)
            HIPCHK(hipMemcpyAsync(d_order, chain.data(), nb * 4, hipMemcpyHostToDevice, st));
            // blocks the reference would have handed out: up to the first one it fails on, and that
            // fit (in block order: start pointer, then room for the block's bytes, then its last run).
            // The reference stops at the first block whose CRC fails, so a damaged block that decodes
            // to something huge must not turn into a capacity error of a later one: the CRCs of the
            // blocks before the stop are compared first, below.
            // Output offsets inside one pass are 32-bit: a chain that expands to more than that (zeros:
            // ~46 MB per -9 block) is written in several passes over [k0, k1).
            int rc2 = BZ2HIP_OK;
            const char *msg2 = "";
            u32 k0 = 0;
            size_t bi = 0;
            // true = the stream that ends before block k of the chain fails its CRC (state set for the return)
            auto at_bound = [&](u32 k, size_t out_before) -> bool {
                if (bi >= bounds.size() || bounds[bi].k != k) return false;
                if (bounds[bi].stored != stream_crc) {
                    *produced = out_before;
                    if (blocks_out) *blocks_out = nblocks + (k - k0);
                    c->err = "BZip2 stream CRC error";
                    return true;
                }
                stream_crc = 0;
                nstreams++;
                if (streams_out) *streams_out = nstreams;
                c->dec_consumed = bounds[bi].consumed;
                bi++;
                return false;
            };
            while (k0 < nb && rc2 == BZ2HIP_OK) {
                std::vector<u32> h_base(1, 0);
                u64 total = 0;
                u32 maxlen = 0, k1 = k0;
                for (u32 k = k0; k < nb; k++) {
                    if (k == first_bad_ptr) { rc2 = BZ2HIP_E_START_PTR; msg2 = "BZip2 start pointer invalid"; break; }
                    if (out_total + total + h_outlen[k] > cap || (h_flags[k] & 2u)) {
                        rc2 = BZ2HIP_E_CAP;
                        msg2 = "output buffer too small";
                        break;
                    }
                    if (k > k0 && total + h_outlen[k] > c->dec_pass_bytes) break; // next pass
                    if (h_flags[k] & 1u) { rc2 = BZ2HIP_E_FORMAT; msg2 = "block ends where a repeat count is due"; break; }
                    total += h_outlen[k];
                    maxlen = std::max(maxlen, h_outlen[k]);
                    h_base.push_back((u32)total);
                    k1 = k + 1;
                }
                const u32 good = k1 - k0; // blocks [k0, k1) are expanded in this pass
                if (good) {
                    u8 *o = d_out + out_total;
                    HIPCHK(hipMemcpyAsync(d_outbase, h_base.data(), (good + 1) * 4, hipMemcpyHostToDevice, st));
                    kd_unrle_write<<<dim3(KD_SEGS, good), 64, 0, st>>>(d_tt, d_n, S, d_segstart, d_segin, d_outbase, o,
                                                                       d_order + k0);
                    KCHECK();
                    const u32 ETPB = (maxlen + 4095) / 4096 + 1;
                    if (ETPB > ETPB_MAX) throw HipError(BZ2HIP_E_HIP, "internal: decoded block longer than 52 x block size");
                    run_crc_blocks(c, o, d_outbase, (int)good, maxlen, d_cagg, ETPB, d_crc);
                    std::vector<u32> got(good);
                    HIPCHK(hipMemcpyAsync(got.data(), d_crc, good * 4, hipMemcpyDeviceToHost, st));
                    HIPCHK(hipStreamSynchronize(st));
                    for (u32 k = 0; k < good; k++) {
                        if (at_bound(k0 + k, out_total + h_base[k])) return BZ2HIP_E_STREAM_CRC;
                        if (got[k] != h_crc[k0 + k]) {
                            // the block's bytes are out before its CRC is compared (S/BZip2InputStream.swift:209-212)
                            *produced = out_total + h_base[k + 1];
                            if (blocks_out) *blocks_out = nblocks + k + 1;
                            c->err = "BZip2 block CRC error";
                            return BZ2HIP_E_BLOCK_CRC;
                        }
                        stream_crc = ((stream_crc << 1) | (stream_crc >> 31)) ^ h_crc[k0 + k];
                    }
                }
                out_total += (size_t)total;
                nblocks += good;
                *produced = out_total;
                if (blocks_out) *blocks_out = nblocks;
                k0 = k1;
                // a pass that stopped at the first block of a stream: the stream before it is complete,
                // and its CRC is compared before that block's error is reported
                if (rc2 != BZ2HIP_OK && at_bound(k1, out_total)) return BZ2HIP_E_STREAM_CRC;
                if (!good) break;
            }
            if (rc2 != BZ2HIP_OK) { c->err = msg2; return rc2; }
            // a stream that ended with the last block of the chain (the block after it is the one that
            // failed to decode, below)
            if (at_bound(nb, out_total)) return BZ2HIP_E_STREAM_CRC;
        }
        if (rc_huff != BZ2HIP_OK) { c->err = "block could not be decoded"; return rc_huff; }
    }
    // not a block: end marker or garbage
    std::vector<u8> tail(16, 0);
    const u64 byte0 = cur >> 3;
    const size_t avail = byte0 < nbytes ? std::min<size_t>(16, nbytes - byte0) : 0;
    if (avail) {
        HIPCHK(hipMemcpyAsync(tail.data(), d_in + byte0, avail, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    tail.resize(avail);
    u32 m1 = 0, m2 = 0, stored = 0;
    if (!host_bits(tail, byte0 * 8, total_bits, cur, 24, &m1) ||
        !host_bits(tail, byte0 * 8, total_bits, cur + 24, 24, &m2)) {
        c->err = "Insufficient data";
        return BZ2HIP_E_INSUFFICIENT;
    }
    if (m1 == 0x177245u && m2 == 0x385090u) {
        if (!host_bits(tail, byte0 * 8, total_bits, cur + 48, 32, &stored)) {
            c->err = "Insufficient data";
            return BZ2HIP_E_INSUFFICIENT;
        }
        if (stored != stream_crc) {
            c->err = "BZip2 stream CRC error";
            return BZ2HIP_E_STREAM_CRC;
        }
        c->dec_consumed = (size_t)((cur + 80 + 7) / 8); // the trailer is followed by zero padding to a byte
        nstreams++;
        if (streams_out) *streams_out = nstreams;
        if (!multi) return BZ2HIP_OK;
        // another stream of the same level right behind this one?  (its header lies within the 16
        // bytes fetched for the trailer: the trailer ends at most 11 bytes into them)
        const size_t nb0 = c->dec_consumed;
        if (nb0 + 4 > nbytes) return BZ2HIP_OK;
        const u8 *h = tail.data() + (nb0 - byte0);
        if ((nb0 - byte0) + 4 > tail.size() || h[0] != 'B' || h[1] != 'Z' || h[2] != 'h' || (int)h[3] - '0' != level)
            return BZ2HIP_OK; // trailing bytes that are no such stream: the caller sees them at dec_consumed
        cur = (u64)nb0 * 8u + 32u;
        stream_crc = 0;
        continue;
    }
    c->err = "BZip2 stream format error";
    return BZ2HIP_E_FORMAT;
    } // streams
}
