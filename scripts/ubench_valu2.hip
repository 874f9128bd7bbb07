// ubench_valu2.hip -- per-instruction issue cost table (wall-clock based, 8 waves/SIMD, all CUs busy).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define KERNEL(NAME, ASMSTR)                                                                                       \
    __global__ __launch_bounds__(256) void NAME(int iters, uint32_t seed, uint32_t *sink, unsigned long long *clk) { \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        uint32_t q = threadIdx.x * 2654435761u + 1, w = q * 7;                                                     \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();               \
        for (int it = 0; it < iters; it++) {                                                                       \
            _Pragma("unroll") for (int u = 0; u < 8; u++)                                                          \
                asm volatile(ASMSTR : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q), "v"(w) : "vcc"); \
        }                                                                                                          \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();               \
        sink[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                              \
        if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }                           \
    }

KERNEL(k_xor, "v_xor_b32 %0, %8, %0\n v_xor_b32 %1, %8, %1\n v_xor_b32 %2, %8, %2\n v_xor_b32 %3, %8, %3\n v_xor_b32 %4, %8, %4\n v_xor_b32 %5, %8, %5\n v_xor_b32 %6, %8, %6\n v_xor_b32 %7, %8, %7\n")
KERNEL(k_and, "v_and_b32 %0, %8, %0\n v_and_b32 %1, %8, %1\n v_and_b32 %2, %8, %2\n v_and_b32 %3, %8, %3\n v_and_b32 %4, %8, %4\n v_and_b32 %5, %8, %5\n v_and_b32 %6, %8, %6\n v_and_b32 %7, %8, %7\n")
KERNEL(k_or, "v_or_b32 %0, %8, %0\n v_or_b32 %1, %8, %1\n v_or_b32 %2, %8, %2\n v_or_b32 %3, %8, %3\n v_or_b32 %4, %8, %4\n v_or_b32 %5, %8, %5\n v_or_b32 %6, %8, %6\n v_or_b32 %7, %8, %7\n")
KERNEL(k_add, "v_add_u32 %0, %8, %0\n v_add_u32 %1, %8, %1\n v_add_u32 %2, %8, %2\n v_add_u32 %3, %8, %3\n v_add_u32 %4, %8, %4\n v_add_u32 %5, %8, %5\n v_add_u32 %6, %8, %6\n v_add_u32 %7, %8, %7\n")
KERNEL(k_mov, "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n")
KERNEL(k_min, "v_min_u32 %0, %8, %0\n v_min_u32 %1, %8, %1\n v_min_u32 %2, %8, %2\n v_min_u32 %3, %8, %3\n v_min_u32 %4, %8, %4\n v_min_u32 %5, %8, %5\n v_min_u32 %6, %8, %6\n v_min_u32 %7, %8, %7\n")
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %8, %1\n v_bcnt_u32_b32 %2, %8, %2\n v_bcnt_u32_b32 %3, %8, %3\n v_bcnt_u32_b32 %4, %8, %4\n v_bcnt_u32_b32 %5, %8, %5\n v_bcnt_u32_b32 %6, %8, %6\n v_bcnt_u32_b32 %7, %8, %7\n")
KERNEL(k_bcnt0, "v_bcnt_u32_b32 %0, %0, 0\n v_bcnt_u32_b32 %1, %1, 0\n v_bcnt_u32_b32 %2, %2, 0\n v_bcnt_u32_b32 %3, %3, 0\n v_bcnt_u32_b32 %4, %4, 0\n v_bcnt_u32_b32 %5, %5, 0\n v_bcnt_u32_b32 %6, %6, 0\n v_bcnt_u32_b32 %7, %7, 0\n")
KERNEL(k_cmp, "v_cmp_lt_u32 vcc, %8, %0\n v_cmp_lt_u32 vcc, %8, %1\n v_cmp_lt_u32 vcc, %8, %2\n v_cmp_lt_u32 vcc, %8, %3\n v_cmp_lt_u32 vcc, %8, %4\n v_cmp_lt_u32 vcc, %8, %5\n v_cmp_lt_u32 vcc, %8, %6\n v_cmp_lt_u32 vcc, %8, %7\n")
KERNEL(k_pkmin16, "v_pk_min_u16 %0, %8, %0\n v_pk_min_u16 %1, %8, %1\n v_pk_min_u16 %2, %8, %2\n v_pk_min_u16 %3, %8, %3\n v_pk_min_u16 %4, %8, %4\n v_pk_min_u16 %5, %8, %5\n v_pk_min_u16 %6, %8, %6\n v_pk_min_u16 %7, %8, %7\n")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3\n v_lshlrev_b32 %4, 1, %4\n v_lshlrev_b32 %5, 1, %5\n v_lshlrev_b32 %6, 1, %6\n v_lshlrev_b32 %7, 1, %7\n")
KERNEL(k_fma, "v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n")
KERNEL(k_fmac, "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n")
KERNEL(k_xorbcnt, "v_xor_b32 %4, %8, %0\n v_xor_b32 %5, %8, %1\n v_xor_b32 %6, %8, %2\n v_xor_b32 %7, %8, %3\n v_bcnt_u32_b32 %0, %4, %0\n v_bcnt_u32_b32 %1, %5, %1\n v_bcnt_u32_b32 %2, %6, %2\n v_bcnt_u32_b32 %3, %7, %3\n")
KERNEL(k_xor_indep, "v_xor_b32 %0, %8, %9\n v_xor_b32 %1, %8, %9\n v_xor_b32 %2, %8, %9\n v_xor_b32 %3, %8, %9\n v_xor_b32 %4, %8, %9\n v_xor_b32 %5, %8, %9\n v_xor_b32 %6, %8, %9\n v_xor_b32 %7, %8, %9\n")
KERNEL(k_bcnt_indep, "v_bcnt_u32_b32 %0, %8, %9\n v_bcnt_u32_b32 %1, %8, %9\n v_bcnt_u32_b32 %2, %8, %9\n v_bcnt_u32_b32 %3, %8, %9\n v_bcnt_u32_b32 %4, %8, %9\n v_bcnt_u32_b32 %5, %8, %9\n v_bcnt_u32_b32 %6, %8, %9\n v_bcnt_u32_b32 %7, %8, %9\n")

typedef void (*kfn)(int, uint32_t, uint32_t *, unsigned long long *);
int run(const char *name, kfn f, int ncu, uint32_t *sink, unsigned long long *clk) {
    const int iters = 4000;
    for (int bpc : {2, 8}) {
        const int blocks = ncu * bpc;
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, 100, 1u, sink, clk);
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, iters, 3u, sink, clk);
        CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[2]; CHK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
        const double mhz = (double)h[0] / (double)h[1] * 100.0;
        const double inst_per_simd = (double)bpc * iters * 64.0;   // wave-instructions issued on one SIMD
        printf("%-14s waves/SIMD=%d  wall=%.3f ms  clock=%.0f MHz  cycles/wave-inst=%.2f  %.1f T lane-ops/s\n", name, bpc, ms, mhz,
               ms * 1e-3 * mhz * 1e6 / inst_per_simd, (double)blocks * 256 * iters * 64.0 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    uint32_t *sink; unsigned long long *clk;
    CHK(hipMalloc(&sink, p.multiProcessorCount * 8 * 256 * 4)); CHK(hipMalloc(&clk, 16));
    const int ncu = p.multiProcessorCount;
#define R(k) run(#k, k, ncu, sink, clk)
    R(k_xor); R(k_xor_indep); R(k_and); R(k_or); R(k_add); R(k_mov); R(k_lshl); R(k_min); R(k_cmp); R(k_pkmin16);
    R(k_bcnt); R(k_bcnt0); R(k_bcnt_indep); R(k_xorbcnt); R(k_fma); R(k_fmac);
    return 0;
}
