// ubench_valu5.hip -- issue cost of the instructions an FP4 expansion is made of: shifts against x + x, v_and_or_b32 against v_and + v_or, 64-bit forms
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71"
#define KERNEL(NAME, BODY)                                                                                   \
    __global__ __launch_bounds__(256) void NAME(int iters, uint32_t seed, uint32_t *sink, unsigned long long *clk) { \
        asm volatile("s_mov_b32 s20, 0x88888888" ::: "s20"); unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();         \
        for (int it = 0; it < iters; it++) {                                                                 \
            _Pragma("unroll") for (int u = 0; u < 8; u++) asm volatile(BODY ::: CLOB, "s20", "vcc");                        \
        }                                                                                                    \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();         \
        uint32_t r; asm volatile("v_xor_b32 %0, v56, v57\n v_xor_b32 %0, %0, v58\n v_xor_b32 %0, %0, v59\n v_xor_b32 %0, %0, v60\n v_xor_b32 %0, %0, v61\n v_xor_b32 %0, %0, v62\n v_xor_b32 %0, %0, v63" : "=v"(r)); \
        sink[blockIdx.x * 256 + threadIdx.x] = r;                                                            \
        if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }                     \
    }
KERNEL(k_lshl32, "v_lshlrev_b32 v56, 1, v40\n v_lshlrev_b32 v57, 1, v42\n v_lshlrev_b32 v58, 1, v44\n v_lshlrev_b32 v59, 1, v46\n v_lshlrev_b32 v60, 1, v48\n v_lshlrev_b32 v61, 1, v50\n v_lshlrev_b32 v62, 1, v52\n v_lshlrev_b32 v63, 1, v54\n ")
KERNEL(k_add_self, "v_add_u32 v56, v40, v40\n v_add_u32 v57, v42, v42\n v_add_u32 v58, v44, v44\n v_add_u32 v59, v46, v46\n v_add_u32 v60, v48, v48\n v_add_u32 v61, v50, v50\n v_add_u32 v62, v52, v52\n v_add_u32 v63, v54, v54\n ")
KERNEL(k_lshl64, "v_lshlrev_b64 v[56:57], 1, v[40:41]\n v_lshlrev_b64 v[58:59], 1, v[42:43]\n v_lshlrev_b64 v[60:61], 1, v[44:45]\n v_lshlrev_b64 v[62:63], 1, v[46:47]\n v_lshlrev_b64 v[56:57], 1, v[48:49]\n v_lshlrev_b64 v[58:59], 1, v[50:51]\n v_lshlrev_b64 v[60:61], 1, v[52:53]\n v_lshlrev_b64 v[62:63], 1, v[54:55]\n ")
KERNEL(k_and_or, "v_and_or_b32 v56, v40, s20, v41\n v_and_or_b32 v57, v42, s20, v43\n v_and_or_b32 v58, v44, s20, v45\n v_and_or_b32 v59, v46, s20, v47\n v_and_or_b32 v60, v48, s20, v49\n v_and_or_b32 v61, v50, s20, v51\n v_and_or_b32 v62, v52, s20, v53\n v_and_or_b32 v63, v54, s20, v55\n ")
KERNEL(k_and_then_or, "v_and_b32 v56, s20, v40\n v_or_b32 v56, v56, v41\n v_and_b32 v57, s20, v42\n v_or_b32 v57, v57, v43\n v_and_b32 v58, s20, v44\n v_or_b32 v58, v58, v45\n v_and_b32 v59, s20, v46\n v_or_b32 v59, v59, v47\n ")
KERNEL(k_bfi, "v_bfi_b32 v56, s20, v40, v41\n v_bfi_b32 v57, s20, v42, v43\n v_bfi_b32 v58, s20, v44, v45\n v_bfi_b32 v59, s20, v46, v47\n v_bfi_b32 v60, s20, v48, v49\n v_bfi_b32 v61, s20, v50, v51\n v_bfi_b32 v62, s20, v52, v53\n v_bfi_b32 v63, s20, v54, v55\n ")
KERNEL(k_mov64, "v_mov_b64 v[56:57], v[40:41]\n v_mov_b64 v[58:59], v[42:43]\n v_mov_b64 v[60:61], v[44:45]\n v_mov_b64 v[62:63], v[46:47]\n v_mov_b64 v[56:57], v[48:49]\n v_mov_b64 v[58:59], v[50:51]\n v_mov_b64 v[60:61], v[52:53]\n v_mov_b64 v[62:63], v[54:55]\n ")
KERNEL(k_max3_i32, "v_max3_i32 v56, v40, v41, v56\n v_max3_i32 v57, v42, v43, v57\n v_max3_i32 v58, v44, v45, v58\n v_max3_i32 v59, v46, v47, v59\n v_max3_i32 v60, v48, v49, v60\n v_max3_i32 v61, v50, v51, v61\n v_max3_i32 v62, v52, v53, v62\n v_max3_i32 v63, v54, v55, v63\n ")
KERNEL(k_cndmask, "v_cndmask_b32 v56, v40, v41, vcc\n v_cndmask_b32 v57, v42, v43, vcc\n v_cndmask_b32 v58, v44, v45, vcc\n v_cndmask_b32 v59, v46, v47, vcc\n v_cndmask_b32 v60, v48, v49, vcc\n v_cndmask_b32 v61, v50, v51, vcc\n v_cndmask_b32 v62, v52, v53, vcc\n v_cndmask_b32 v63, v54, v55, vcc\n ")
KERNEL(k_pk_add_u16, "v_pk_add_u16 v56, v40, v40\n v_pk_add_u16 v57, v42, v42\n v_pk_add_u16 v58, v44, v44\n v_pk_add_u16 v59, v46, v46\n v_pk_add_u16 v60, v48, v48\n v_pk_add_u16 v61, v50, v50\n v_pk_add_u16 v62, v52, v52\n v_pk_add_u16 v63, v54, v54\n ")
KERNEL(k_lshl_add, "v_lshl_add_u32 v56, v40, 1, v41\n v_lshl_add_u32 v57, v42, 1, v43\n v_lshl_add_u32 v58, v44, 1, v45\n v_lshl_add_u32 v59, v46, 1, v47\n v_lshl_add_u32 v60, v48, 1, v49\n v_lshl_add_u32 v61, v50, 1, v51\n v_lshl_add_u32 v62, v52, 1, v53\n v_lshl_add_u32 v63, v54, 1, v55\n ")
KERNEL(k_mix_add_andor, "v_add_u32 v56, v40, v40\n v_and_or_b32 v60, v56, s20, v41\n v_add_u32 v57, v42, v42\n v_and_or_b32 v61, v57, s20, v43\n v_add_u32 v58, v44, v44\n v_and_or_b32 v62, v58, s20, v45\n v_add_u32 v59, v46, v46\n v_and_or_b32 v63, v59, s20, v47\n ")
KERNEL(k_mix_lshl_andor, "v_lshlrev_b32 v56, 1, v40\n v_and_or_b32 v60, v56, s20, v41\n v_lshlrev_b32 v57, 1, v42\n v_and_or_b32 v61, v57, s20, v43\n v_lshlrev_b32 v58, 1, v44\n v_and_or_b32 v62, v58, s20, v45\n v_lshlrev_b32 v59, 1, v46\n v_and_or_b32 v63, v59, s20, v47\n ")

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
    R(k_lshl32); R(k_add_self); R(k_lshl64); R(k_and_or); R(k_and_then_or); R(k_bfi); R(k_mov64); R(k_max3_i32); R(k_cndmask); R(k_pk_add_u16); R(k_lshl_add); R(k_mix_add_andor); R(k_mix_lshl_andor);
    return 0;
}
