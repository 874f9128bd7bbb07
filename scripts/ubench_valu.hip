// ubench_valu.hip -- issue-rate micro-benchmarks for the popcount roofline.
// For each instruction mix: cycles per wave-instruction per SIMD (s_memtime) at 1..8 waves/SIMD,
// plus the shader clock (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MIX>
__global__ __launch_bounds__(256) void k(int iters, uint32_t seed, uint32_t *sink, unsigned long long *cyc, unsigned long long *rt) {
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    const uint32_t q = threadIdx.x * 2654435761u + 1;
    __builtin_amdgcn_s_barrier();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (MIX == 0) {  // v_xor_b32 only (VOP2)
                asm volatile("v_xor_b32 %0, %8, %0\n v_xor_b32 %1, %8, %1\n v_xor_b32 %2, %8, %2\n v_xor_b32 %3, %8, %3\n"
                             "v_xor_b32 %4, %8, %4\n v_xor_b32 %5, %8, %5\n v_xor_b32 %6, %8, %6\n v_xor_b32 %7, %8, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q));
            } else if (MIX == 1) {  // v_bcnt_u32_b32 only (VOP3, accumulate)
                asm volatile("v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %8, %1\n v_bcnt_u32_b32 %2, %8, %2\n v_bcnt_u32_b32 %3, %8, %3\n"
                             "v_bcnt_u32_b32 %4, %8, %4\n v_bcnt_u32_b32 %5, %8, %5\n v_bcnt_u32_b32 %6, %8, %6\n v_bcnt_u32_b32 %7, %8, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q));
            } else if (MIX == 2) {  // xor + bcnt pairs, 4 independent chains
                asm volatile("v_xor_b32 %4, %8, %0\n v_xor_b32 %5, %8, %1\n v_xor_b32 %6, %8, %2\n v_xor_b32 %7, %8, %3\n"
                             "v_bcnt_u32_b32 %0, %4, %0\n v_bcnt_u32_b32 %1, %5, %1\n v_bcnt_u32_b32 %2, %6, %2\n v_bcnt_u32_b32 %3, %7, %3\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q));
            } else if (MIX == 3) {  // v_min_u32 (VOP2)
                asm volatile("v_min_u32 %0, %8, %0\n v_min_u32 %1, %8, %1\n v_min_u32 %2, %8, %2\n v_min_u32 %3, %8, %3\n"
                             "v_min_u32 %4, %8, %4\n v_min_u32 %5, %8, %5\n v_min_u32 %6, %8, %6\n v_min_u32 %7, %8, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q));
            } else if (MIX == 4) {  // v_med3_u32 (VOP3)
                asm volatile("v_med3_u32 %0, %8, %0, %1\n v_med3_u32 %1, %8, %1, %2\n v_med3_u32 %2, %8, %2, %3\n v_med3_u32 %3, %8, %3, %4\n"
                             "v_med3_u32 %4, %8, %4, %5\n v_med3_u32 %5, %8, %5, %6\n v_med3_u32 %6, %8, %6, %7\n v_med3_u32 %7, %8, %7, %0\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q));
            } else if (MIX == 5) {  // v_lshl_or_b32 (VOP3)
                asm volatile("v_lshl_or_b32 %0, %0, 1, %8\n v_lshl_or_b32 %1, %1, 1, %8\n v_lshl_or_b32 %2, %2, 1, %8\n v_lshl_or_b32 %3, %3, 1, %8\n"
                             "v_lshl_or_b32 %4, %4, 1, %8\n v_lshl_or_b32 %5, %5, 1, %8\n v_lshl_or_b32 %6, %6, 1, %8\n v_lshl_or_b32 %7, %7, 1, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q));
            } else if (MIX == 6) {  // v_fma_f32 reference
                asm volatile("v_fma_f32 %0, %8, %0, %0\n v_fma_f32 %1, %8, %1, %1\n v_fma_f32 %2, %8, %2, %2\n v_fma_f32 %3, %8, %3, %3\n"
                             "v_fma_f32 %4, %8, %4, %4\n v_fma_f32 %5, %8, %5, %5\n v_fma_f32 %6, %8, %6, %6\n v_fma_f32 %7, %8, %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q));
            } else if (MIX == 7) {  // v_pk_fma_f32 reference
                asm volatile("v_pk_fma_f32 %0, %4, %0, %0\n v_pk_fma_f32 %1, %4, %1, %1\n v_pk_fma_f32 %2, %4, %2, %2\n v_pk_fma_f32 %3, %4, %3, %3\n"
                             "v_pk_fma_f32 %0, %4, %0, %0\n v_pk_fma_f32 %1, %4, %1, %1\n v_pk_fma_f32 %2, %4, %2, %2\n v_pk_fma_f32 %3, %4, %3, %3\n"
                             : "+v"(*(uint64_t*)&a0), "+v"(*(uint64_t*)&a2), "+v"(*(uint64_t*)&a4), "+v"(*(uint64_t*)&a6) : "v"((uint64_t)q));
            } else if (MIX == 8) {  // v_xor with SGPR operand
                asm volatile("v_xor_b32 %0, %8, %0\n v_xor_b32 %1, %8, %1\n v_xor_b32 %2, %8, %2\n v_xor_b32 %3, %8, %3\n"
                             "v_xor_b32 %4, %8, %4\n v_xor_b32 %5, %8, %5\n v_xor_b32 %6, %8, %6\n v_xor_b32 %7, %8, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(seed));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    sink[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}

template <int MIX>
int run(const char *name, int ncu) {
    uint32_t *sink; unsigned long long *cyc, *rt;
    const int maxblocks = ncu * 8;
    CHK(hipMalloc(&sink, maxblocks * 256 * 4)); CHK(hipMalloc(&cyc, maxblocks * 8)); CHK(hipMalloc(&rt, maxblocks * 8));
    const int iters = 4000;
    for (int bpc : {1, 2, 4, 8}) {  // 256-thread blocks per CU = waves per SIMD
        const int blocks = ncu * bpc;
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<MIX>, dim3(blocks), dim3(256), 0, 0, 100, 1u, sink, cyc, rt);
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k<MIX>, dim3(blocks), dim3(256), 0, 0, iters, 3u, sink, cyc, rt);
        CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> hc(blocks), hr(blocks);
        CHK(hipMemcpy(hc.data(), cyc, blocks * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(hr.data(), rt, blocks * 8, hipMemcpyDeviceToHost));
        std::sort(hc.begin(), hc.end()); std::sort(hr.begin(), hr.end());
        const double med_c = (double)hc[blocks / 2], med_r = (double)hr[blocks / 2];
        const double insts_per_wave = (double)iters * 64.0;  // 8 unrolled x 8 instructions
        // one SIMD hosts bpc waves of this kernel: cycles per wave-instruction per SIMD = cycles / (insts * bpc)
        const double lane_ops = (double)blocks * 256.0 * insts_per_wave * (MIX == 7 ? 2.0 : 1.0);
        printf("%-14s waves/SIMD=%d  cyc/inst/SIMD=%.2f  clock=%.0f MHz  wall=%.3f ms  %.2f T lane-ops/s\n", name, bpc,
               med_c / (insts_per_wave * bpc), med_c / med_r * 100.0, ms, lane_ops / (ms * 1e-3) / 1e12);
    }
    hipFree(sink); hipFree(cyc); hipFree(rt);
    return 0;
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    printf("%s CUs=%d clockRate=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    const int ncu = p.multiProcessorCount;
    run<0>("v_xor_b32", ncu); run<8>("v_xor_b32(sgpr)", ncu); run<1>("v_bcnt_u32_b32", ncu); run<2>("xor+bcnt", ncu);
    run<3>("v_min_u32", ncu); run<4>("v_med3_u32", ncu); run<5>("v_lshl_or_b32", ncu); run<6>("v_fma_f32", ncu); run<7>("v_pk_fma_f32", ncu);
    return 0;
}
