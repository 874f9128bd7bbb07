// distinct.hip -- MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:363-449), batched over map points: for each
// point the N descriptors of its observations, all-pairs Hamming distances, per-row median (element
// (int)(0.5*(N-1)) of the sorted row, self-distance 0 included) and the first row with the least median.
//
// One wavefront per point.  The descriptors sit in LDS; for row i the wave reads descriptor i (broadcast) and lane l
// computes d(i, l), d(i, l + 64), ... (at most DIST_MAX_N / 64 values per lane, in registers).  The median needs no
// sort: distances are integers in [0, 256], so the k-th smallest is found by a 9-step bisection on the value with a
// ballot + popcount per step.  A point with more than DIST_MAX_N observations (the reference has no limit) does not fit
// this form: its wave streams the rows from global memory instead and finds each row's median in a 257-bin histogram
// in LDS -- slow (N^2 / 64 row loads) and rare.
#include <limits.h>
#include "orbm_internal.h"

namespace {

constexpr int DIST_MAX_N = 512;          // observations per map point that the register form holds; above: the histogram form
constexpr int DIST_PER_LANE = DIST_MAX_N / 64;
constexpr int DIST_WAVES = 4;

__global__ __launch_bounds__(64 * DIST_WAVES) void distinctive_kernel(int n_points, const int32_t *__restrict__ off,
                                                                      const uint32_t *__restrict__ desc,
                                                                      int32_t *__restrict__ best_idx) {
    extern __shared__ __align__(16) uint32_t lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = blockIdx.x * DIST_WAVES + wave;
    if (p >= n_points) return;
    const int b = off[p], N = off[p + 1] - b;
    if (N <= 0) { if (lane == 0) best_idx[p] = -1; return; }                              // :384-385 / :405-406
    uint4 *l_desc = reinterpret_cast<uint4 *>(lds) + (size_t)wave * (2 * DIST_MAX_N);
    const uint4 *g = reinterpret_cast<const uint4 *>(desc) + 2 * (size_t)b;
    if (N > DIST_MAX_N) {                                                                  // wave-uniform
        int *hist = reinterpret_cast<int *>(l_desc);                                      // 320 bins (257 used)
        const int k = (int)(0.5 * (double)(N - 1));                                       // :434
        int bestMedian = INT_MAX, bestI = 0;
        for (int i = 0; i < N; i++) {
            for (int t = lane; t < 320; t += 64) hist[t] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint4 lo = g[2 * i], hi = g[2 * i + 1];
            const uint32_t q[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            for (int j = lane; j < N; j += 64) {
                const uint4 clo = g[2 * j], chi = g[2 * j + 1];
                const uint32_t c[8] = {clo.x, clo.y, clo.z, clo.w, chi.x, chi.y, chi.z, chi.w};
                atomicAdd(&hist[hamming256(q, c)], 1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // lane l owns bins 5 l .. 5 l + 4; the median is the first value whose running count exceeds k
            int c5[5], sum = 0;
#pragma unroll
            for (int u = 0; u < 5; u++) { c5[u] = hist[5 * lane + u]; sum += c5[u]; }
            int incl = sum;
#pragma unroll
            for (int dlt = 1; dlt < 64; dlt <<= 1) {
                const int up = __shfl_up(incl, dlt, 64);
                if (lane >= dlt) incl += up;
            }
            int med = INT_MAX, run = incl - sum;
            if (run <= k && k < incl) {
#pragma unroll
                for (int u = 0; u < 5; u++) { run += c5[u]; if (run > k && med == INT_MAX) med = 5 * lane + u; }
            }
#pragma unroll
            for (int offs = 32; offs > 0; offs >>= 1) med = min(med, __shfl_xor(med, offs, 64));
            if (med < bestMedian) { bestMedian = med; bestI = i; }                        // :436-440 (strict <: first wins)
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) best_idx[p] = bestI;
        return;
    }
    for (int i = lane; i < 2 * N; i += 64) l_desc[i] = g[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // this lane's columns
    uint32_t mine[DIST_PER_LANE][8];
    const int iters = (N + 63) >> 6;
#pragma unroll
    for (int t = 0; t < DIST_PER_LANE; t++) {
        const int j = t * 64 + lane;
        uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
        if (t < iters && j < N) { lo = l_desc[2 * j]; hi = l_desc[2 * j + 1]; }
        mine[t][0] = lo.x; mine[t][1] = lo.y; mine[t][2] = lo.z; mine[t][3] = lo.w;
        mine[t][4] = hi.x; mine[t][5] = hi.y; mine[t][6] = hi.z; mine[t][7] = hi.w;
    }
    const int k = (int)(0.5 * (double)(N - 1));                                           // :434
    int bestMedian = INT_MAX, bestI = 0;                                                 // :426-427
    for (int i = 0; i < N; i++) {
        const uint4 lo = l_desc[2 * i], hi = l_desc[2 * i + 1];
        const uint32_t q[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        uint32_t d[DIST_PER_LANE];
#pragma unroll
        for (int t = 0; t < DIST_PER_LANE; t++) d[t] = (t < iters && t * 64 + lane < N) ? hamming256(q, mine[t]) : 0xFFFFu;
        // smallest v with #(d <= v) > k
        int lo_v = 0, hi_v = 256;
        while (lo_v < hi_v) {
            const int mid = (lo_v + hi_v) >> 1;
            int cnt = 0;
#pragma unroll
            for (int t = 0; t < DIST_PER_LANE; t++)
                if (t < iters) cnt += __popcll(__ballot(d[t] <= (uint32_t)mid));
            if (cnt > k) hi_v = mid; else lo_v = mid + 1;
        }
        if (lo_v < bestMedian) { bestMedian = lo_v; bestI = i; }                          // :436-440 (strict <: first wins)
    }
    if (lane == 0) best_idx[p] = bestI;
}

}  // namespace

extern "C" int orbm_distinctive_descriptors_batch(orbm_ctx *ctx, int n_points, const int32_t *off, const uint8_t *desc,
                                                  int32_t *best_idx, int io_on_device) {
    if (!ctx) return ORBM_EINVAL;
    if (n_points < 0 || (n_points > 0 && (!off || !best_idx)))
        return orbm_fail(ctx, ORBM_EINVAL, "distinctive_descriptors_batch: null argument or negative size");
    if (n_points == 0) return ORBM_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    size_t total = 0;
    if (!io_on_device) {
        if (off[0] != 0) return orbm_fail(ctx, ORBM_EINVAL, "distinctive_descriptors_batch: off[0] must be 0");
        for (int p = 0; p < n_points; p++) {
            const int n = off[p + 1] - off[p];
            if (n < 0) return orbm_fail(ctx, ORBM_EINVAL, "distinctive_descriptors_batch: off not monotone at %d", p);
        }
        total = (size_t)off[n_points];
        if (total > 0 && !desc) return orbm_fail(ctx, ORBM_EINVAL, "distinctive_descriptors_batch: null descriptors");
        int rc = orbm_ws_reserve(ctx, (size_t)(n_points + 1) * 4 + total * 32 + (size_t)n_points * 4 + 4096);
        if (rc) return rc;
    }
    IoStage io(ctx, io_on_device);
    const void *doff, *ddesc = desc; void *dout;
    int rc;
    if ((rc = io.in(off, (size_t)(n_points + 1) * 4, &doff)) || (total && (rc = io.in(desc, total * 32, &ddesc))) ||
        (rc = io.out(best_idx, (size_t)n_points * 4, &dout)))
        return rc;
    const size_t lds = (size_t)DIST_WAVES * DIST_MAX_N * 32;
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&distinctive_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(distinctive_kernel, dim3((n_points + DIST_WAVES - 1) / DIST_WAVES), dim3(64 * DIST_WAVES), lds, ctx->stream,
                       n_points, (const int32_t *)doff, (const uint32_t *)ddesc, (int32_t *)dout);
    HIPCHK(ctx, hipGetLastError());
    return io.finish();
}
