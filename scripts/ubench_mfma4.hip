// ubench_mfma4.hip -- what the chip sustains on back-to-back v_mfma_scale_f32_32x32x64_f8f6f4 (FP4 operands) with
// nothing else going on: the ceiling the brute-force kernel's MFMA roofline fraction should be read against.
// Operands stay in registers; PATTERN 0 = zeros, 1 = random +-1.0 FP4 values (what bf2nn multiplies).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k(int iters, int pattern, float *sink, unsigned long long *clk) {
    uint32_t h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = a;
    if (pattern) {
        for (int i = 0; i < 4; i++) {
            h = h * 1664525u + 1013904223u; a[i] = (int)(0x22222222u | (h & 0x88888888u));
            h = h * 1664525u + 1013904223u; b[i] = (int)(0x22222222u | (h & 0x88888888u));
        }
    }
    v16f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const int one = 0x7F7F7F7F;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, one, 0, one);
        c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 4, 4, 0, one, 0, one);
        c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 4, 4, 0, one, 0, one);
        c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 4, 4, 0, one, 0, one);
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
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; wgs_per_cu++)
        for (int pattern = 0; pattern < 2; pattern++) {
            const int iters = 200000, grid = 256 * wgs_per_cu;
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, 2000, pattern, sink, clk);   // warm
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, iters, pattern, sink, clk);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            CHK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
            const double mfma = (double)grid * 4 * iters * 4;          // wave instructions
            const double tflops = mfma * 2.0 * 32 * 32 * 64 / (ms * 1e-3) / 1e12;
            const double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;   // s_memrealtime ticks at 100 MHz
            printf("%d waves/SIMD, %s: %.1f ms, %.0f TFLOP/s dense FP4 (%.3f of 10066), shader clock %.2f GHz, %.1f cycles per MFMA per SIMD\n",
                   wgs_per_cu, pattern ? "+-1 operands" : "zero operands", ms, tflops, tflops / 10066.3, ghz,
                   (double)h[0] / (4.0 * iters * wgs_per_cu));
        }
    return 0;
}
