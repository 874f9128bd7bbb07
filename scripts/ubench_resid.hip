// ubench_resid.hip -- how many workgroups of an MFMA loop like bf2nn's are really resident per CU, and where the
// dispatcher puts them: every workgroup records its CU (HW_ID / XCC_ID) and its start and end time.
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -o ubench_resid ubench_resid.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <map>
#include <algorithm>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int LDS_V4>
__global__ __launch_bounds__(256) void k(int iters, float *sink, unsigned long long *rec) {
    __shared__ v4i tile[LDS_V4];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    for (int i = threadIdx.x; i < LDS_V4; i += 256) {
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
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rt = 0; rt < 4; rt++) {
            v8i a[4];
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const int r = (((it * 4 + rt) & 7) * 32 + col) & (LDS_V4 / 8 - 1);
                v4i x = tile[r * 8 + ((2 * s + hh) ^ ((r >> 1) & 7))];
                asm volatile("" : "+v"(x));
                a[s] = (v8i){x[0], x[1], x[2], x[3], 0, 0, 0, 0};
            }
            v16f c0 = {0}, c1 = {0};
#pragma unroll
            for (int s = 0; s < 4; s++) {
                c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[s], bq[0][s], c0, 4, 4, 0, one, 0, one);
                c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[s], bq[1][s], c1, 4, 4, 0, one, 0, one);
            }
            float m0 = best, m1 = best;
#pragma unroll
            for (int i = 0; i < 16; i += 2) { m0 = __builtin_fmaxf(__builtin_fmaxf(m0, c0[i]), c0[i + 1]); m1 = __builtin_fmaxf(__builtin_fmaxf(m1, c1[i]), c1[i + 1]); }
            best = __builtin_fmaxf(best, __builtin_fminf(m0, m1) * 1e-9f);
        }
        __syncthreads();
    }
    if (best == 123456.f) sink[0] = best;
    if (threadIdx.x == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        rec[3 * blockIdx.x] = ((unsigned long long)(xcc & 0xF) << 32) | hw;
        rec[3 * blockIdx.x + 1] = t0;
        rec[3 * blockIdx.x + 2] = t1;
    }
}

template <int LDS_V4>
int run(int wgs_per_cu, float *sink) {
    const int grid = 256 * wgs_per_cu, iters = 3000;
    unsigned long long *rec;
    CHK(hipMalloc(&rec, (size_t)grid * 24));
    hipLaunchKernelGGL(k<LDS_V4>, dim3(grid), dim3(256), 0, 0, 50, sink, rec);
    CHK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k<LDS_V4>, dim3(grid), dim3(256), 0, 0, iters, sink, rec);
    CHK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)grid * 3);
    CHK(hipMemcpy(h.data(), rec, (size_t)grid * 24, hipMemcpyDeviceToHost));
    unsigned long long tmin = ~0ull, tmax = 0;
    std::map<uint32_t, std::vector<std::pair<unsigned long long, int>>> ev;     // per CU: (time, +1 / -1)
    double dur = 0;
    for (int b = 0; b < grid; b++) {
        const uint32_t hw = (uint32_t)h[3 * b], xcc = (uint32_t)(h[3 * b] >> 32);
        const uint32_t cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);   // xcc, se, sh, cu
        ev[cu].push_back({h[3 * b + 1], +1});
        ev[cu].push_back({h[3 * b + 2], -1});
        tmin = std::min(tmin, h[3 * b + 1]); tmax = std::max(tmax, h[3 * b + 2]);
        dur += (double)(h[3 * b + 2] - h[3 * b + 1]);
    }
    int peak_max = 0; double peak_sum = 0; size_t most = 0, least = ~(size_t)0;
    for (auto &kv : ev) {
        auto &v = kv.second;
        std::sort(v.begin(), v.end());
        int cur = 0, peak = 0;
        for (auto &e : v) { cur += e.second; peak = std::max(peak, cur); }
        peak_max = std::max(peak_max, peak); peak_sum += peak;
        most = std::max(most, v.size() / 2); least = std::min(least, v.size() / 2);
    }
    printf("LDS %3d KB, %d workgroups per CU in the grid: %zu CUs seen, workgroups per CU %zu..%zu, peak resident per CU %d (mean of peaks %.2f), "
           "kernel %.2f ms, mean workgroup %.2f ms\n", LDS_V4 / 64, wgs_per_cu, ev.size(), least, most, peak_max, peak_sum / ev.size(),
           (double)(tmax - tmin) / 1e5, dur / grid / 1e5);
    CHK(hipFree(rec));
    return 0;
}

int main() {
    float *sink;
    CHK(hipMalloc(&sink, 4));
    for (int w : {1, 2, 4, 5, 8})
        if (run<2048>(w, sink)) return 1;
    if (run<512>(5, sink) || run<512>(8, sink)) return 1;
    return 0;
}
