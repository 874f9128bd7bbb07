// ubench_valu4.hip -- does the 2-cycle issue of v_xor_b32 depend on VGPR banks / operand reuse?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71"
#define KERNEL(NAME, BODY)                                                                                   \
    __global__ __launch_bounds__(256) void NAME(int iters, uint32_t seed, uint32_t *sink, unsigned long long *clk) { \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();         \
        for (int it = 0; it < iters; it++) {                                                                 \
            _Pragma("unroll") for (int u = 0; u < 8; u++) asm volatile(BODY ::: CLOB);                        \
        }                                                                                                    \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();         \
        uint32_t r; asm volatile("v_xor_b32 %0, v56, v57\n v_xor_b32 %0, %0, v58\n v_xor_b32 %0, %0, v59\n v_xor_b32 %0, %0, v60\n v_xor_b32 %0, %0, v61\n v_xor_b32 %0, %0, v62\n v_xor_b32 %0, %0, v63" : "=v"(r)); \
        sink[blockIdx.x * 256 + threadIdx.x] = r;                                                            \
        if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }                     \
    }
// sources v40..v55 (16 regs), destinations v56..v63
KERNEL(k_diffbank, "v_xor_b32 v56, v40, v41\n v_xor_b32 v57, v42, v43\n v_xor_b32 v58, v44, v45\n v_xor_b32 v59, v46, v47\n v_xor_b32 v60, v48, v49\n v_xor_b32 v61, v50, v51\n v_xor_b32 v62, v52, v53\n v_xor_b32 v63, v54, v55\n")
KERNEL(k_samebank, "v_xor_b32 v56, v40, v44\n v_xor_b32 v57, v41, v45\n v_xor_b32 v58, v42, v46\n v_xor_b32 v59, v43, v47\n v_xor_b32 v60, v48, v52\n v_xor_b32 v61, v49, v53\n v_xor_b32 v62, v50, v54\n v_xor_b32 v63, v51, v55\n")
KERNEL(k_samesrc, "v_xor_b32 v56, v40, v41\n v_xor_b32 v57, v40, v41\n v_xor_b32 v58, v40, v41\n v_xor_b32 v59, v40, v41\n v_xor_b32 v60, v40, v41\n v_xor_b32 v61, v40, v41\n v_xor_b32 v62, v40, v41\n v_xor_b32 v63, v40, v41\n")
KERNEL(k_onecommon, "v_xor_b32 v56, v40, v41\n v_xor_b32 v57, v40, v42\n v_xor_b32 v58, v40, v43\n v_xor_b32 v59, v40, v44\n v_xor_b32 v60, v40, v45\n v_xor_b32 v61, v40, v46\n v_xor_b32 v62, v40, v47\n v_xor_b32 v63, v40, v48\n")
KERNEL(k_inplace, "v_xor_b32 v56, v40, v56\n v_xor_b32 v57, v41, v57\n v_xor_b32 v58, v42, v58\n v_xor_b32 v59, v43, v59\n v_xor_b32 v60, v44, v60\n v_xor_b32 v61, v45, v61\n v_xor_b32 v62, v46, v62\n v_xor_b32 v63, v47, v63\n")
KERNEL(k_mix_diff, "v_xor_b32 v56, v40, v41\n v_bcnt_u32_b32 v64, v56, v64\n v_xor_b32 v57, v42, v43\n v_bcnt_u32_b32 v64, v57, v64\n v_xor_b32 v58, v44, v45\n v_bcnt_u32_b32 v64, v58, v64\n v_xor_b32 v59, v46, v47\n v_bcnt_u32_b32 v64, v59, v64\n")
KERNEL(k_mix_common, "v_xor_b32 v56, v40, v41\n v_bcnt_u32_b32 v64, v56, v64\n v_xor_b32 v57, v40, v42\n v_bcnt_u32_b32 v65, v57, v65\n v_xor_b32 v58, v40, v43\n v_bcnt_u32_b32 v66, v58, v66\n v_xor_b32 v59, v40, v44\n v_bcnt_u32_b32 v67, v59, v67\n")
KERNEL(k_xor4_bcnt4, "v_xor_b32 v56, v40, v41\n v_xor_b32 v57, v42, v43\n v_xor_b32 v58, v44, v45\n v_xor_b32 v59, v46, v47\n v_bcnt_u32_b32 v64, v56, v64\n v_bcnt_u32_b32 v64, v57, v64\n v_bcnt_u32_b32 v64, v58, v64\n v_bcnt_u32_b32 v64, v59, v64\n")
KERNEL(k_bcnt_only, "v_bcnt_u32_b32 v64, v56, v64\n v_bcnt_u32_b32 v64, v57, v64\n v_bcnt_u32_b32 v64, v58, v64\n v_bcnt_u32_b32 v64, v59, v64\n v_bcnt_u32_b32 v65, v56, v65\n v_bcnt_u32_b32 v65, v57, v65\n v_bcnt_u32_b32 v65, v58, v65\n v_bcnt_u32_b32 v65, v59, v65\n")

typedef void (*kfn)(int, uint32_t, uint32_t *, unsigned long long *);
int run(const char *name, kfn f, int ncu, uint32_t *sink, unsigned long long *clk) {
    const int iters = 4000, bpc = 8;
    const int blocks = ncu * bpc;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, 100, 1u, sink, clk);
    CHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, iters, 3u, sink, clk);
    CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CHK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double mhz = (double)h[0] / (double)h[1] * 100.0;
    printf("%-14s wall=%.3f ms  clock=%.0f MHz  cycles/wave-inst=%.2f\n", name, ms, mhz, ms * 1e-3 * mhz * 1e6 / ((double)bpc * iters * 64.0));
    return 0;
}
int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    uint32_t *sink; unsigned long long *clk;
    CHK(hipMalloc(&sink, p.multiProcessorCount * 8 * 256 * 4)); CHK(hipMalloc(&clk, 16));
    const int ncu = p.multiProcessorCount;
#define R(k) run(#k, k, ncu, sink, clk)
    R(k_diffbank); R(k_samebank); R(k_samesrc); R(k_onecommon); R(k_inplace); R(k_mix_diff); R(k_mix_common); R(k_xor4_bcnt4); R(k_bcnt_only);
    return 0;
}
