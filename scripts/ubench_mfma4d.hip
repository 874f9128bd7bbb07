// ubench_mfma4d.hip -- which part of the brute-force kernel's block loop keeps the matrix pipe from being busy?
// The loop of bf2nn_mfma4r_kernel<.,4,2> without memory: per block 16 products in four groups that share one candidate
// fragment, the fragment of the next group made by vector instructions in between, then the fold of 4 x 16 results
// (8 v_min3 each) and a branch that is never taken.  Sign-free operand values as in the kernel.  Modes (bit mask):
//   1  the fragment is made between the groups, into the registers the previous group read (what hipcc emits for the kernel)
//   2  the fold and its branch
//   4  with 1: the fragment of the NEXT group is made before the current group's products (second register set)
//   8  s_setprio 1 around the products
// Three workgroups of 256 threads per CU = three waves per SIMD, as the kernel has.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ v8i expand(uint32_t w) {
    return (v8i){(int)(~w & 0x11111111u), (int)(~w & 0x22222222u), (int)(~w & 0x44444444u), (int)(~(w >> 1) & 0x44444444u), 0, 0, 0, 0};
}

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void k(int iters, int *sink, unsigned long long *clk) {
    uint32_t h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    v8i bq[4][4];
    for (int t = 0; t < 4; t++)
        for (int s = 0; s < 4; s++) {
            bq[t][s] = (v8i){0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 0; i < 4; i++) { h = h * 1664525u + 1013904223u; bq[t][s][i] = (int)((h ^ (h >> 13)) & 0x44444444u); }
        }
    uint32_t w[4];
    for (int i = 0; i < 4; i++) { h = h * 1664525u + 1013904223u; w[i] = h ^ (h >> 11); }
    int T = 0x7F000000, hits = 0;
    asm volatile("" : "+v"(T));
    v16f cinit;
    for (int i = 0; i < 16; i++) cinit[i] = (float)(100 + ((h >> i) & 31));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        if (MODE & 8) __builtin_amdgcn_s_setprio(1);
        v16f acc[4];
        v8i a = (MODE & 1) ? expand(w[0]) : bq[0][0];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            v8i an = a;
            if ((MODE & 5) == 5 && s < 3) an = expand(w[s + 1]);
#pragma unroll
            for (int t = 0; t < 4; t++)
                acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bq[t][s], s == 0 ? cinit : acc[t], 4, 4, 0, 0, 0, 0);
            if ((MODE & 5) == 1 && s < 3) an = expand(w[s + 1]);
            a = an;
        }
        if (MODE & 8) __builtin_amdgcn_s_setprio(0);
        if (MODE & 2) {
            bool fire = false;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                int m = min(min(__float_as_int(acc[t][0]), __float_as_int(acc[t][1])), T);
#pragma unroll
                for (int r = 2; r < 16; r += 2) m = min(min(m, __float_as_int(acc[t][r])), __float_as_int(acc[t][r + 1]));
                fire = fire || m != T;
            }
            if (__any(fire)) { hits++; T = T - 1; }
        } else {
#pragma unroll
            for (int t = 0; t < 4; t++) asm volatile("" ::"v"(acc[t]));
        }
        // the next block's rows: something the compiler cannot hoist
#pragma unroll
        for (int i = 0; i < 4; i++) w[i] = w[i] * 1664525u + (uint32_t)it;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (hits == 123456789) sink[0] = hits;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE>
int run(const char *name, int *sink, unsigned long long *clk, hipEvent_t e0, hipEvent_t e1) {
    unsigned long long h[2];
    const int iters = 60000, grid = 256 * 3;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, 500, sink, clk);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, iters, sink, clk);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    CHK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double mfma = (double)grid * 4 * iters * 16;
    const double tflops = mfma * 2.0 * 32 * 32 * 64 / (ms * 1e-3) / 1e12;
    const double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
    // busy = matrix-pipe cycles (32 per product) over the SIMD-cycles of the run at the clock it held
    const double busy = mfma * 32 / (1024.0 * ghz * 1e9 * ms * 1e-3);
    printf("mode %2d %-58s %.1f ms, %.3f of 10066 TFLOP/s, clock %.2f GHz, matrix pipe busy %.3f\n", MODE, name, ms, tflops / 10066.3, ghz, busy);
    fflush(stdout);
    return 0;
}

int main() {
    int *sink; unsigned long long *clk;
    CHK(hipMalloc(&sink, 4)); CHK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; rep++) {
        if (run<0>("products only", sink, clk, e0, e1)) return 1;
        if (run<8>("products only, setprio", sink, clk, e0, e1)) return 1;
        if (run<1>("+ fragment between the groups (in place)", sink, clk, e0, e1)) return 1;
        if (run<5>("+ fragment one group ahead (second set)", sink, clk, e0, e1)) return 1;
        if (run<2>("+ fold", sink, clk, e0, e1)) return 1;
        if (run<3>("+ fragment in place + fold", sink, clk, e0, e1)) return 1;
        if (run<11>("+ fragment in place + fold, setprio (the kernel)", sink, clk, e0, e1)) return 1;
        if (run<7>("+ fragment ahead + fold", sink, clk, e0, e1)) return 1;
        if (run<15>("+ fragment ahead + fold, setprio", sink, clk, e0, e1)) return 1;
    }
    return 0;
}
