// common.hpp -- shared device helpers and the batch workspace of libbz2hip (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

#define WAVE 64

// ---------------------------------------------------------------- wave / block primitives
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// v_writelane_b32: lane `l` of `old` := the uniform value `v` (the compiler's intrinsic, which this
// clang has no builtin for; it routes the lane select through M0 and places the wait states)
extern "C" __device__ int bz2_writelane_i32(int, int, int) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ u32 write_lane(u32 old, u32 v, int l) {
    return (u32)bz2_writelane_i32((int)v, l, (int)old);
}

__device__ __forceinline__ u32 wave_incl_scan_add(u32 v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u32 t = __shfl_up(v, d, 64);
        if (lane_id() >= d) v += t;
    }
    return v;
}
__device__ __forceinline__ u32 wave_incl_scan_max(u32 v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u32 t = __shfl_up(v, d, 64);
        if (lane_id() >= d) v = max(v, t);
    }
    return v;
}
__device__ __forceinline__ u32 wave_reduce_add(u32 v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ u32 wave_reduce_max(u32 v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
    return v;
}

// Block-wide exclusive add scan of one value per thread (blockDim.x = NT, multiple of 64).
// sm must hold NT/64 + 1 words.  Returns the exclusive prefix; *total = block sum.
template <int NT>
__device__ __forceinline__ u32 block_excl_scan_add(u32 v, u32 *sm, u32 *total) {
    constexpr int NW = NT / 64;
    u32 inc = wave_incl_scan_add(v);
    __syncthreads();
    if (lane_id() == 63) sm[wave_id()] = inc;
    __syncthreads();
    u32 base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        u32 t = sm[w];
        if (w < wave_id()) base += t;
        tot += t;
    }
    *total = tot;
    return base + inc - v;
}
// Block-wide inclusive max scan.
template <int NT>
__device__ __forceinline__ u32 block_incl_scan_max(u32 v, u32 *sm, u32 *total) {
    constexpr int NW = NT / 64;
    u32 inc = wave_incl_scan_max(v);
    __syncthreads();
    if (lane_id() == 63) sm[wave_id()] = inc;
    __syncthreads();
    u32 base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        u32 t = sm[w];
        if (w < wave_id()) base = max(base, t);
        tot = max(tot, t);
    }
    *total = tot;
    return max(base, inc);
}
template <int NT>
__device__ __forceinline__ u32 block_reduce_add(u32 v, u32 *sm) {
    constexpr int NW = NT / 64;
    v = wave_reduce_add(v);
    __syncthreads();
    if (lane_id() == 0) sm[wave_id()] = v;
    __syncthreads();
    u32 tot = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) tot += sm[w];
    return tot;
}
template <int NT>
__device__ __forceinline__ u32 block_reduce_max(u32 v, u32 *sm) {
    constexpr int NW = NT / 64;
    v = wave_reduce_max(v);
    __syncthreads();
    if (lane_id() == 0) sm[wave_id()] = v;
    __syncthreads();
    u32 tot = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) tot = max(tot, sm[w]);
    return tot;
}

// ---------------------------------------------------------------- profiling classes
enum ProfClass {
    P_RLE1_SCAN = 0,
    P_RLE1_SPLIT,
    P_RLE1_EMIT,
    P_CRC,
    P_BWT_INIT,
    P_RADIX_HIST,
    P_RADIX_SCAN,
    P_RADIX_SCATTER,
    P_BWT_KEYS,
    P_BWT_SSORT,
    P_BWT_RERANK,
    P_BWT_EMIT,
    P_BWT_TIE,
    P_MTF_STATE,
    P_MTF_COMPOSE,
    P_MTF_REPLAY,
    P_RLE2,
    P_HUF_SEED,
    P_HUF_SELECT,
    P_HUF_LENGTHS,
    P_HUF_CODES,
    P_HUF_HEADER,
    P_PACK,
    P_FINAL,
    P_COUNT
};

// ---------------------------------------------------------------- constants
#define RTILE 4096          // radix / rerank tile (elements)
#define RT_THREADS 256
#define RT_ITEMS 16
#define MTF_CHUNK 1024      // symbols per MTF chunk (one wave)
#define R2TILE 4096         // RLE2 tile
#define PK_TILE 2048        // bit-pack tile (symbols)
#define GROUP 50
#define MAX_ALPHA 258
#define MAX_TABLES 6
#define SEL_MAX 18002       // selectors per block (ceil(900001/50) = 18001)
#define SEL_STRIDE 18048
#define HS_GROUPS_PER_WG 1024
#define HS_PART_MAX 18      // ceil(18001/1024)
