// Single-wave issue rates on gfx950: dependent SALU chain, dependent VALU chain, VALU->SGPR->SALU
// round trips (v_readlane then s_add), taken branches.  (The s_add asm statements declare the
// SCC clobber: without it the loop branch reads the carry of the asm and never ends.)  Prints ticks (s_memtime) and wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void probe(unsigned *out, long long *ticks, int n, int waves_busy) {
    const int lane = threadIdx.x & 63;
    unsigned s = out[0];
    unsigned v = lane;
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s) : : "scc");
    }
    long long t1 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) asm volatile("v_add_u32 %0, %0, 1" : "+v"(v));
    }
    long long t2 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            unsigned r;
            asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(r) : "v"(v));
            asm volatile("s_add_u32 %0, %0, %1" : "+s"(s) : "s"(r) : "scc");
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "s"(s));
        }
    }
    long long t3 = clock64();
    unsigned acc = 0;
#pragma unroll 1
    for (int i = 0; i < n * 16; i++) { // one taken branch + 3 SALU per iteration
        asm volatile("s_add_u32 %0, %0, 1" : "+s"(acc) : : "scc");
    }
    long long t4 = clock64();
    // four independent SALU chains, interleaved; four independent VALU chains
    unsigned s1 = s, s2 = s + 1, s3 = s + 2, s4 = s + 3;
#pragma unroll 1
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            asm volatile("s_add_u32 %0, %0, 1" : "+s"(s1) : : "scc");
            asm volatile("s_add_u32 %0, %0, 1" : "+s"(s2) : : "scc");
            asm volatile("s_add_u32 %0, %0, 1" : "+s"(s3) : : "scc");
            asm volatile("s_add_u32 %0, %0, 1" : "+s"(s4) : : "scc");
        }
    }
    long long t5 = clock64();
    unsigned v1 = v, v2 = v + 1, v3 = v + 2, v4 = v + 3;
#pragma unroll 1
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            asm volatile("v_add_u32 %0, %0, 1" : "+v"(v1));
            asm volatile("v_add_u32 %0, %0, 1" : "+v"(v2));
            asm volatile("v_add_u32 %0, %0, 1" : "+v"(v3));
            asm volatile("v_add_u32 %0, %0, 1" : "+v"(v4));
        }
    }
    long long t6 = clock64();
    // SALU and VALU alternating, independent of each other
#pragma unroll 1
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            asm volatile("s_add_u32 %0, %0, 1" : "+s"(s1) : : "scc");
            asm volatile("v_add_u32 %0, %0, 1" : "+v"(v1));
        }
    }
    long long t7 = clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ticks[0] = t1 - t0; ticks[1] = t2 - t1; ticks[2] = t3 - t2; ticks[3] = t4 - t3;
        ticks[4] = t5 - t4; ticks[5] = t6 - t5; ticks[6] = t7 - t6;
    }
    out[1 + blockIdx.x * blockDim.x + threadIdx.x] = s + v + acc + s1 + s2 + s3 + s4 + v1 + v2 + v3 + v4;
}
int main() {
    unsigned *d; long long *t;
    hipMalloc(&d, 1 << 24); hipMalloc(&t, 128);
    hipMemset(d, 0, 1 << 24);
    const int n = 20000;
    for (int grid : {1, 256, 1024}) {
        for (int threads : {64, 256}) {
            probe<<<grid, threads>>>(d, t, n, 0);
            hipDeviceSynchronize();
            auto a = std::chrono::steady_clock::now();
            probe<<<grid, threads>>>(d, t, n, 0);
            hipDeviceSynchronize();
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
            long long h[7];
            hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
            long long tot = h[0] + h[1] + h[2] + h[3];
            printf("grid %4d x %3d: wall %.2f ms | per instr: salu %.2f valu %.2f readlane+salu+valu triple %.2f loop-iter(branch) %.2f | 4 independent chains: salu %.2f valu %.2f | salu/valu alternating %.2f\n",
                   grid, threads, ms, h[0] / (16.0 * n), h[1] / (16.0 * n), h[2] / (8.0 * n), h[3] / (16.0 * n),
                   h[4] / (16.0 * n), h[5] / (16.0 * n), h[6] / (16.0 * n));
            (void)tot;
        }
    }
    return 0;
}
