// ubench_mfma4b.hip -- which ingredient of bf2nn_mfma4_kernel's inner loop keeps the matrix cores from 100 %.
// MODE bits: 1 = A fragments read from LDS (4 x ds_read_b128 per 8 MFMAs), 2 = the v_max3 fold of both result tiles,
// 4 = one s_barrier per 32 MFMAs, 8 = staging of the next 128 candidate rows (16 KB) per pass with global_load_lds,
// 16 = staged from a 64 KB region instead of a 16 MB one, 64 = no vmcnt wait before the barrier, 128 = staging through
// registers, 256 = one staging load per row tile instead of all at the top, 512 = raw bits + expansion, 1024 = lanes read consecutive chunks, 2048 = scalar base + 32-bit lane offset,
// 4096 = a fifth wave per workgroup issues all sixteen staging loads of a pass and multiplies nothing (see main).
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -o ubench_mfma4b ubench_mfma4b.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int MODE, int LDSV4 = 2048, int NLD = 4>
__global__ __launch_bounds__(320) void k(int iters, float *sink, unsigned long long *clk, const v4i *__restrict__ store) {
    __shared__ v4i tile[LDSV4];                                 // 32 KB like the product kernel
    uint32_t h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    for (int i = threadIdx.x; i < LDSV4; i += blockDim.x) {
        v4i v;
        for (int j = 0; j < 4; j++) { h = h * 1664525u + 1013904223u; v[j] = (int)(0x22222222u | (h & 0x88888888u)); }
        tile[i] = v;
    }
    v8i bq[2][4];
    for (int t = 0; t < 2; t++)
        for (int s = 0; s < 4; s++) {
            v8i v = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int j = 0; j < 4; j++) { h = h * 1664525u + 1013904223u; v[j] = (int)(0x22222222u | (h & 0x88888888u)); }
            bq[t][s] = v;
        }
    __syncthreads();
    const int lane = threadIdx.x & 63, col = lane & 31, hh = lane >> 5;
    const int one = 0x7F7F7F7F;
    float best = -1e30f;
    int fired = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const int wave = threadIdx.x >> 6;
    for (int it = 0; it < iters; it++) {
        const uint32_t blk = ((blockIdx.x * 2654435761u) >> 8) + (uint32_t)it * 37u;
        const v4i *src0 = store + (size_t)(blk & ((MODE & 16) ? 3u : 1023u)) * 1024;
        v4i stg[4];
        v4i raw = {0, 0, 0, 0};
        if (MODE & 512) raw = *(const __attribute__((address_space(1))) v4i *)(src0 + threadIdx.x);    // 256 lanes x 16 B = 128 rows x 32 B
        auto stage_part = [&](int j) {
            // the product kernel's staging: 128 rows x 128 B per pass, into the buffer not being read
            const int g0 = __builtin_amdgcn_readfirstlane((wave * 4 + j) * 64);
            const int g = g0 + lane, r = g >> 3, cpos = g & 7;
            const uint32_t blk_off = __builtin_amdgcn_readfirstlane((blk & ((MODE & 16) ? 3u : 1023u)) * 1024u);
            const v4i *src = (MODE & 2048) ? store + (blk_off + (uint32_t)(r * 8 + (cpos ^ ((r >> 1) & 7)))) : src0 + r * 8 + ((MODE & 1024) ? cpos : (cpos ^ ((r >> 1) & 7)));
            if (MODE & 128) stg[j] = *(const __attribute__((address_space(1))) v4i *)src;      // to registers, written to LDS at the end of the pass
            else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                  (__attribute__((address_space(3))) void *)(tile + ((it + 1) & 1) * 1024 + g0), 16, 0, 0);
        };
        if ((MODE & 8) && (MODE & 4096)) {
            if (wave == 4) {
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const int g0 = __builtin_amdgcn_readfirstlane(j * 64);
                    const int g = g0 + lane, r = g >> 3, cpos = g & 7;
                    const v4i *src = src0 + r * 8 + (cpos ^ ((r >> 1) & 7));
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(tile + ((it + 1) & 1) * 1024 + g0), 16, 0, 0);
                }
            }
        } else if ((MODE & 8) && !(MODE & 256) && !(MODE & 512)) {
#pragma unroll
            for (int j = 0; j < NLD; j++) stage_part(j & 3);
        }
#pragma unroll
        for (int rt = 0; rt < 4; rt++) {                        // 4 row tiles = one stage of the product kernel
            if ((MODE & 4096) && wave == 4) break;
            if ((MODE & 8) && (MODE & 256)) stage_part(rt);        // spread: one part per row tile
            v8i a[4];
#pragma unroll
            for (int s = 0; s < 4; s++) {
                v4i x;
                if (MODE & 1) {
                    const int r = (((MODE & 8) ? (it & 1) * 128 + rt * 32 : ((it * 4 + rt) & 7) * 32) + col) & (LDSV4 / 8 - 1);
                    x = tile[r * 8 + ((2 * s + hh) ^ ((r >> 1) & 7))];
                } else x = tile[0];                             // hoisted by the compiler: no LDS traffic in the loop
                asm volatile("" : "+v"(x));                      // a new value as far as the compiler knows: nothing is hoisted
                a[s] = (v8i){x[0], x[1], x[2], x[3], 0, 0, 0, 0};
            }
            v16f c0 = {0}, c1 = {0};
#pragma unroll
            for (int s = 0; s < 4; s++) {
                c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[s], bq[0][s], c0, 4, 4, 0, one, 0, one);
                c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[s], bq[1][s], c1, 4, 4, 0, one, 0, one);
            }
            if (MODE & 2) {
                float m0 = best, m1 = best;
#pragma unroll
                for (int i = 0; i < 16; i += 2) { m0 = __builtin_fmaxf(__builtin_fmaxf(m0, c0[i]), c0[i + 1]); m1 = __builtin_fmaxf(__builtin_fmaxf(m1, c1[i]), c1[i + 1]); }
                if (__any(m0 > 300.f || m1 > 300.f)) fired++;
                best = __builtin_fmaxf(best, __builtin_fminf(m0, m1) * 1e-9f);
            } else {
                best += c0[0] + c1[5];
            }
        }
        if (MODE & 512) {
            // bits -> FP4 +-1.0: two bits select one byte of {0x22, 0x2A, 0xA2, 0xAA}
            const int r = threadIdx.x >> 1, hh = threadIdx.x & 1;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const uint32_t w = (uint32_t)raw[d];
                v4i out;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    uint32_t x = (w >> (8 * b)) & 0xFFu;
                    x = x | (x << 6); x = x | (x << 12);
                    out[b] = (int)__builtin_amdgcn_perm(0u, 0xAAA22A22u, x & 0x03030303u);
                }
                const int c = 4 * hh + d;
                tile[((it + 1) & 1) * 1024 + r * 8 + (c ^ ((r >> 1) & 7))] = out;
            }
        }
        if ((MODE & 8) && (MODE & 128)) {
#pragma unroll
            for (int j = 0; j < 4; j++) tile[((it + 1) & 1) * 1024 + (wave * 4 + j) * 64 + lane] = stg[j];
        }
        if ((MODE & 8) && !(MODE & 64) && !(MODE & 128)) __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): this wave's part of the next stage has landed
        if (MODE & 4) __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (best == 123456.f || fired == 77) sink[0] = best;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE, int LDSV4 = 2048, int NLD = 4>
int run(int wgs_per_cu, float *sink, unsigned long long *clk, const v4i *store) {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    unsigned long long h[2];
    const int iters = 12000, grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL((k<MODE, LDSV4, NLD>), dim3(grid), dim3((MODE & 4096) ? 320 : 256), 0, 0, 200, sink, clk, store);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, LDSV4, NLD>), dim3(grid), dim3((MODE & 4096) ? 320 : 256), 0, 0, iters, sink, clk, store);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    CHK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double mfma = (double)grid * 4 * iters * 32;
    const double tflops = mfma * 2.0 * 32 * 32 * 64 / (ms * 1e-3) / 1e12;
    const double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
    printf("%d loads/pass, LDS %d KB, mode %d%s%s%s (%s%s%s), %d waves/SIMD: %.1f ms, %.0f TFLOP/s (%.3f of 10066), clock %.2f GHz, %.1f cycles per MFMA per SIMD\n", NLD, LDSV4 / 64, MODE, (MODE & 64) ? " nowait" : "", (MODE & 128) ? " via-registers" : "", (MODE & 4096) ? " loader-wave" : (MODE & 2048) ? " saddr" : (MODE & 1024) ? " linear-global" : (MODE & 512) ? " raw-bits+expand" : (MODE & 256) ? " spread" : "",
           (MODE & 1) ? "lds " : "", (MODE & 2) ? "fold " : "", (MODE & 4) ? ((MODE & 8) ? ((MODE & 16) ? ((MODE & 32) ? "barrier staging-small-half" : "barrier staging-small") : ((MODE & 32) ? "barrier staging-half" : "barrier staging")) : "barrier") : "", wgs_per_cu, ms, tflops, tflops / 10066.3, ghz,
           (double)h[0] / (32.0 * iters * wgs_per_cu));
    return 0;
}

int main() {
    float *sink; unsigned long long *clk;
    CHK(hipMalloc(&sink, 4)); CHK(hipMalloc(&clk, 16));
    v4i *store;
    CHK(hipMalloc(&store, (size_t)16 << 20)); CHK(hipMemset(store, 0x2A, (size_t)16 << 20));
    {
        int nb = 0;
        hipFuncAttributes fa;
        CHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k<15, 2048, 4>, 256, 0));
        CHK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k<15, 2048, 4>)));
        printf("runtime: %d workgroups of 256 threads per CU for the staging loop (%d VGPRs, %zu B static LDS)\n", nb, fa.numRegs, (size_t)fa.sharedSizeBytes);
        hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
        printf("device: %d CUs, %zu B LDS per workgroup max, %zu B per CU, %d regs per block, max threads per CU %d\n", prop.multiProcessorCount,
               (size_t)prop.sharedMemPerBlock, (size_t)prop.sharedMemPerMultiprocessor, prop.regsPerBlock, prop.maxThreadsPerMultiProcessor);
    }
    const int w = 4;     // workgroups per CU in the grid = what is resident with 32 KB of LDS each (scripts/ubench_resid.hip): one round, no tail
    if (run<0, 2048, 0>(w, sink, clk, store) ||            // MFMAs on register operands only (2 accumulator chains per wave)
        run<3, 2048, 0>(w, sink, clk, store) ||            // + A fragments from LDS + the max3 fold
        run<7, 2048, 0>(w, sink, clk, store) ||            // + one barrier per 32 MFMAs
        run<15, 2048, 1>(w, sink, clk, store) ||           // + staging, 1 / 2 / 4 global_load_lds (1 KB each) per wave and pass
        run<15, 2048, 2>(w, sink, clk, store) ||
        run<15, 2048, 4>(w, sink, clk, store) ||           //   = the product kernel's inner loop
        run<31, 2048, 4>(w, sink, clk, store) ||           //   the same from a 64 KB region (L2 hits)
        run<15 + 128, 2048, 4>(w, sink, clk, store) ||     //   through registers + ds_write_b128 instead
        run<7 + 512, 2048, 0>(w, sink, clk, store) ||      //   raw descriptor bits (1 load) expanded to FP4 by the vector ALU
        run<15 + 1024, 2048, 4>(w, sink, clk, store) ||    //   lanes read consecutive 16-byte chunks (no swizzle on the global side)
        run<15 + 2048, 2048, 4>(w, sink, clk, store) ||    //   scalar base + 32-bit lane offset addressing
        run<15 + 4096, 2048, 0>(w, sink, clk, store) ||    //   a fifth wave does all the staging
        run<7, 2048, 0>(w, sink, clk, store) || run<15, 2048, 4>(w, sink, clk, store) || run<15 + 4096, 2048, 0>(w, sink, clk, store))
        return 1;
    return 0;
}
