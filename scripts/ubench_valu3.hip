// ubench_valu3.hip -- cost of ONE 256-bit comparison (8 xor + 8 bcnt) under different instruction orders.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// P=1: xor,bcnt adjacent (dependent)  P=2: 8 xor then 8 bcnt  P=3: two comparisons interleaved  P=4: xor of cmp k+1 between bcnts of cmp k
template <int P>
__global__ __launch_bounds__(256) void k(int iters, uint32_t seed, uint32_t *sink, unsigned long long *clk) {
    uint32_t q0 = seed + threadIdx.x, q1 = q0 * 3, q2 = q0 * 5, q3 = q0 * 7, q4 = q0 * 11, q5 = q0 * 13, q6 = q0 * 17, q7 = q0 * 19;
    uint32_t c0 = q0 * 23, c1 = q0 * 29, c2 = q0 * 31, c3 = q0 * 37, c4 = q0 * 41, c5 = q0 * 43, c6 = q0 * 47, c7 = q0 * 53;
    uint32_t acc = 0, acc2 = 0, x0, x1, x2, x3, x4, x5, x6, x7, y0, y1, y2, y3, y4, y5, y6, y7;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (P == 1) {
                asm volatile(
                    "v_xor_b32 %1, %9, %17\n v_bcnt_u32_b32 %0, %1, %0\n v_xor_b32 %2, %10, %18\n v_bcnt_u32_b32 %0, %2, %0\n"
                    "v_xor_b32 %3, %11, %19\n v_bcnt_u32_b32 %0, %3, %0\n v_xor_b32 %4, %12, %20\n v_bcnt_u32_b32 %0, %4, %0\n"
                    "v_xor_b32 %5, %13, %21\n v_bcnt_u32_b32 %0, %5, %0\n v_xor_b32 %6, %14, %22\n v_bcnt_u32_b32 %0, %6, %0\n"
                    "v_xor_b32 %7, %15, %23\n v_bcnt_u32_b32 %0, %7, %0\n v_xor_b32 %8, %16, %24\n v_bcnt_u32_b32 %0, %8, %0\n"
                    : "+v"(acc), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7)
                    : "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(q4), "v"(q5), "v"(q6), "v"(q7), "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(c4), "v"(c5), "v"(c6), "v"(c7));
            } else if (P == 2) {
                asm volatile(
                    "v_xor_b32 %1, %9, %17\n v_xor_b32 %2, %10, %18\n v_xor_b32 %3, %11, %19\n v_xor_b32 %4, %12, %20\n"
                    "v_xor_b32 %5, %13, %21\n v_xor_b32 %6, %14, %22\n v_xor_b32 %7, %15, %23\n v_xor_b32 %8, %16, %24\n"
                    "v_bcnt_u32_b32 %0, %1, %0\n v_bcnt_u32_b32 %0, %2, %0\n v_bcnt_u32_b32 %0, %3, %0\n v_bcnt_u32_b32 %0, %4, %0\n"
                    "v_bcnt_u32_b32 %0, %5, %0\n v_bcnt_u32_b32 %0, %6, %0\n v_bcnt_u32_b32 %0, %7, %0\n v_bcnt_u32_b32 %0, %8, %0\n"
                    : "+v"(acc), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7)
                    : "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(q4), "v"(q5), "v"(q6), "v"(q7), "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(c4), "v"(c5), "v"(c6), "v"(c7));
            } else if (P == 3) {  // half a step: two accumulators, 4 words each -> same op count as one comparison
                asm volatile(
                    "v_xor_b32 %2, %10, %18\n v_xor_b32 %3, %11, %19\n v_bcnt_u32_b32 %0, %2, %0\n v_bcnt_u32_b32 %1, %3, %1\n"
                    "v_xor_b32 %4, %12, %20\n v_xor_b32 %5, %13, %21\n v_bcnt_u32_b32 %0, %4, %0\n v_bcnt_u32_b32 %1, %5, %1\n"
                    "v_xor_b32 %6, %14, %22\n v_xor_b32 %7, %15, %23\n v_bcnt_u32_b32 %0, %6, %0\n v_bcnt_u32_b32 %1, %7, %1\n"
                    "v_xor_b32 %8, %16, %24\n v_xor_b32 %9, %17, %25\n v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %9, %1\n"
                    : "+v"(acc), "+v"(acc2), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7)
                    : "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(q4), "v"(q5), "v"(q6), "v"(q7), "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(c4), "v"(c5), "v"(c6), "v"(c7));
            } else if (P == 4) {  // bcnt of previous xors (y) interleaved with xors of the next comparison (x)
                asm volatile(
                    "v_xor_b32 %1, %17, %25\n v_bcnt_u32_b32 %0, %9, %0\n v_xor_b32 %2, %18, %26\n v_bcnt_u32_b32 %0, %10, %0\n"
                    "v_xor_b32 %3, %19, %27\n v_bcnt_u32_b32 %0, %11, %0\n v_xor_b32 %4, %20, %28\n v_bcnt_u32_b32 %0, %12, %0\n"
                    "v_xor_b32 %5, %21, %29\n v_bcnt_u32_b32 %0, %13, %0\n v_xor_b32 %6, %22, %30\n v_bcnt_u32_b32 %0, %14, %0\n"
                    "v_xor_b32 %7, %23, %31\n v_bcnt_u32_b32 %0, %15, %0\n v_xor_b32 %8, %24, %32\n v_bcnt_u32_b32 %0, %16, %0\n"
                    : "+v"(acc), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7)
                    : "v"(y0), "v"(y1), "v"(y2), "v"(y3), "v"(y4), "v"(y5), "v"(y6), "v"(y7),
                      "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(q4), "v"(q5), "v"(q6), "v"(q7), "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(c4), "v"(c5), "v"(c6), "v"(c7));
                y0 = x0; y1 = x1; y2 = x2; y3 = x3; y4 = x4; y5 = x5; y6 = x6; y7 = x7;
            }
            c0 += acc;  // keep the candidate changing (1 fast op per comparison, not counted)
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    sink[blockIdx.x * 256 + threadIdx.x] = acc + acc2;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

typedef void (*kfn)(int, uint32_t, uint32_t *, unsigned long long *);
int run(const char *name, kfn f, int ncu, uint32_t *sink, unsigned long long *clk) {
    const int iters = 4000;
    for (int bpc : {2, 4, 8}) {
        const int blocks = ncu * bpc;
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, 100, 1u, sink, clk);
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, iters, 3u, sink, clk);
        CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[2]; CHK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
        const double mhz = (double)h[0] / (double)h[1] * 100.0;
        const double cmps_per_simd = (double)bpc * iters * 4.0;   // wave-comparisons on one SIMD
        printf("%-22s waves/SIMD=%d  wall=%.3f ms  clock=%.0f MHz  cycles/wave-cmp=%.1f  %.2f T cmps/s\n", name, bpc, ms, mhz,
               ms * 1e-3 * mhz * 1e6 / cmps_per_simd, (double)blocks * 256 * iters * 4.0 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    uint32_t *sink; unsigned long long *clk;
    CHK(hipMalloc(&sink, p.multiProcessorCount * 8 * 256 * 4)); CHK(hipMalloc(&clk, 16));
    const int ncu = p.multiProcessorCount;
    run("adjacent xor,bcnt", k<1>, ncu, sink, clk); run("8 xor then 8 bcnt", k<2>, ncu, sink, clk);
    run("2 chains interleaved", k<3>, ncu, sink, clk); run("sw-pipelined xor/bcnt", k<4>, ncu, sink, clk);
    return 0;
}
