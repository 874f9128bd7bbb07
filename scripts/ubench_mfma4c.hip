// ubench_mfma4c.hip -- does the VALUE of the FP4 operands change what the chip sustains on back-to-back
// v_mfma_f32_32x32x64_f8f6f4?  ubench_mfma4.hip showed zeros at 2.38 GHz and random +-1 at 2.1 GHz and falling (power).
// The Hamming identity has a second form: candidates as {0, 1}, queries as +-1 gives dot = popcount(q) - d, so half of
// the products are zero.  Patterns (A x B), four distinct operand pairs per loop trip so the operand buses toggle:
//   0 zeros x zeros   1 +-1 x +-1   2 {0,1} x +-1   3 {0,1} x {0,1}   4 {0,1} (1 in 4 set) x +-1
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t rnd(uint32_t &h) { h = h * 1664525u + 1013904223u; return h ^ (h >> 15); }
__device__ __forceinline__ int operand_word(uint32_t &h, int kind) {
    const uint32_t r = rnd(h);
    if (kind == 1) return (int)(0x22222222u | (r & 0x88888888u));      // +-1.0
    if (kind == 2) return (int)(r & 0x22222222u);                        // 0 or 1.0
    if (kind == 3) return (int)(r & rnd(h) & 0x22222222u);               // 0 or 1.0, one in four set
    return 0;
}

__global__ __launch_bounds__(256) void k(int iters, int kind_a, int kind_b, float *sink, unsigned long long *clk) {
    uint32_t h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    v8i a[4], b[4];
    for (int j = 0; j < 4; j++) {
        a[j] = v8i{0, 0, 0, 0, 0, 0, 0, 0}; b[j] = a[j];
        for (int i = 0; i < 4; i++) { a[j][i] = operand_word(h, kind_a); b[j][i] = operand_word(h, kind_b); }
    }
    v16f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[0], b[0], c0, 4, 4, 0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[1], b[1], c1, 4, 4, 0, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[2], b[2], c2, 4, 4, 0, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[3], b[3], c3, 4, 4, 0, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 16; i++) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 123456.f) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main() {
    float *sink; unsigned long long *clk, h[2];
    CHK(hipMalloc(&sink, 4)); CHK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int kinds[5][2] = {{0, 0}, {1, 1}, {2, 1}, {2, 2}, {3, 1}};
    const char *names[5] = {"zeros x zeros", "+-1 x +-1", "{0,1} x +-1", "{0,1} x {0,1}", "{0,1} sparse x +-1"};
    for (int rep = 0; rep < 2; rep++)
        for (int p = 0; p < 5; p++) {
            const int iters = 800000, grid = 256 * 2;   // ~100 ms per launch: long enough for the power limit to act
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, 2000, kinds[p][0], kinds[p][1], sink, clk);   // warm
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, iters, kinds[p][0], kinds[p][1], sink, clk);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            CHK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
            const double mfma = (double)grid * 4 * iters * 4;          // wave instructions
            const double tflops = mfma * 2.0 * 32 * 32 * 64 / (ms * 1e-3) / 1e12;
            const double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;   // s_memrealtime ticks at 100 MHz
            printf("%-20s %.1f ms, %.0f TFLOP/s dense FP4 (%.3f of 10066), shader clock %.2f GHz\n", names[p], ms, tflops,
                   tflops / 10066.3, ghz);
            fflush(stdout);
        }
    return 0;
}
