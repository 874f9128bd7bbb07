// stereo.hip -- descriptor stage of Frame::ComputeStereoMatches (Frame.cc:632-728), batched over stereo frames:
// for every left keypoint the right keypoint with the least Hamming distance among those whose row band covers the
// left keypoint's row (Frame.cc:645-660), whose octave is within +-1 (:712) and whose x lies in [uL - maxD, uL]
// (:690-717); strict '<' against an initial TH_HIGH, candidates in ascending right index (the order the row table
// yields them).  The SAD sub-pixel refinement that follows (:734-833) works on the image pyramid and stays with
// the caller.
//
// The row table is not materialised: membership is two integer compares, so every left keypoint (one per lane)
// scans ALL right keypoints, staged through LDS 256 at a time and read as broadcasts, in ascending index order.
// The gates come first and the distance is only computed when some lane of the wave passes them.  Workgroups of one
// wave (64 left keypoints) when the batch is small -- a single stereo pair, the per-frame case, then spreads over 32
// workgroups -- and of four waves, which share the staging of the right keypoints, when the batch fills the chip.
#include <algorithm>
#include "orbm_internal.h"

namespace {

constexpr int ST_TILE = 256;

template <int ST_THREADS>
__global__ __launch_bounds__(ST_THREADS) void stereo_kernel(const FrameDev *__restrict__ table, const int32_t *__restrict__ fl,
                                                     const int32_t *__restrict__ fr, const float *__restrict__ max_d,
                                                     int blocks_per_pair, int stride, int32_t *__restrict__ best_idx,
                                                     int32_t *__restrict__ best_dist) {
    __shared__ v4u s_desc[2 * ST_TILE];
    __shared__ float s_x[ST_TILE];
    __shared__ int s_minr[ST_TILE], s_maxr[ST_TILE], s_oct[ST_TILE];
    const int p = blockIdx.x / blocks_per_pair;
    const int iL = (blockIdx.x - p * blocks_per_pair) * ST_THREADS + (int)threadIdx.x;
    const FrameG L(table[fl[p]]), R(table[fr[p]]);
    const int tid = threadIdx.x;
    const bool have = iL < L.n;
    uint32_t q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float uL = 0.f, minU = 0.f;
    int row = 0, levelL = 0;
    bool live = false;
    if (have) {
        const gptr<const v4u> g = (gptr<const v4u>)L.desc + 2 * (size_t)iL;
        const v4u lo = g[0], hi = g[1];
        q[0] = lo.x; q[1] = lo.y; q[2] = lo.z; q[3] = lo.w; q[4] = hi.x; q[5] = hi.y; q[6] = hi.z; q[7] = hi.w;
        uL = L.x[iL];
        row = (int)L.y[iL];                                   // vRowIndices[vL]: float -> index truncation (:684)
        levelL = L.octave[iL];
        minU = __fsub_rn(uL, max_d[p]);                       // :690
        live = !(uL < 0);                                     // maxU = uL - 0 (:691-694)
    }
    int bestDist = ORBM_TH_HIGH, bestIdxR = -1;               // :696
    for (int base = 0; base < R.n; base += ST_TILE) {
        __syncthreads();
        for (int t = tid; t < ST_TILE; t += ST_THREADS) {
            const int j = base + t;
            if (j < R.n) {
                const gptr<const v4u> g = (gptr<const v4u>)R.desc + 2 * (size_t)j;
                s_desc[2 * t] = g[0]; s_desc[2 * t + 1] = g[1];
                const float kpY = R.y[j];
                const int o = R.octave[j];
                const float r = __fmul_rn(2.0f, R.sf[o]);         // :654
                s_x[t] = R.x[j];
                s_maxr[t] = (int)ceilf(__fadd_rn(kpY, r));        // :655
                s_minr[t] = (int)floorf(__fsub_rn(kpY, r));       // :656
                s_oct[t] = o;
            }
        }
        __syncthreads();
        const int cnt = min(ST_TILE, R.n - base);
        for (int c = 0; c < cnt; c++) {
            const float uR = s_x[c];
            const int o = s_oct[c];
            const bool ok = live && row >= s_minr[c] && row <= s_maxr[c] &&          // row band (:658-659)
                            !(o < levelL - 1 || o > levelL + 1) &&                     // :712
                            uR >= minU && uR <= uL;                                    // :717
            if (!__any(ok)) continue;                                                  // wave-uniform
            const v4u lo = s_desc[2 * c], hi = s_desc[2 * c + 1];
            const uint32_t cd[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            const int dist = (int)hamming256(q, cd);
            if (ok && dist < bestDist) { bestDist = dist; bestIdxR = base + c; }      // :722-726
        }
    }
    if (iL < stride) {                                        // rows beyond the frame's keypoints: no match
        best_idx[(size_t)p * stride + iL] = bestIdxR;
        best_dist[(size_t)p * stride + iL] = bestDist;
    }
}

// ---- the same search with the right keypoints binned by the first row of their band -------------------------------------
// Every workgroup first counting-sorts the right keypoints in LDS by min row (2048 bins, clamped), keeping per entry
// {index | octave << 16, min row, max row, x}.  A left keypoint on row v then only visits the bins [v - W, v], W = the
// tallest band any right keypoint can have (ceil(4 * the largest scale factor) + 2): a few dozen entries instead of all
// of them, each tested exactly (min row <= v <= max row, octave, disparity range).  The scan order inside a bin is
// arbitrary, so the choice is by key (distance << 16 | index) over the candidates below TH_HIGH: least distance, lowest
// index among equals = the reference's strict '<' over ascending indices.  Right frames above 4096 keypoints use the
// scan form above.
constexpr int ST_BINS = 2048;
constexpr int ST_SORT_MAX_N = 4096;

template <int ST_THREADS>
__global__ __launch_bounds__(ST_THREADS) void stereo_sorted_kernel(const FrameDev *__restrict__ table, const int32_t *__restrict__ fl,
                                                                   const int32_t *__restrict__ fr, const float *__restrict__ max_d,
                                                                   int blocks_per_pair, int stride, int32_t *__restrict__ best_idx,
                                                                   int32_t *__restrict__ best_dist) {
    extern __shared__ __align__(16) uint4 s_entry[];                    // R.n entries, then ST_BINS + 1 bin starts
    __shared__ uint32_t s_cursor[ST_BINS];
    const int p = blockIdx.x / blocks_per_pair;
    const int iL = (blockIdx.x - p * blocks_per_pair) * ST_THREADS + (int)threadIdx.x;
    const FrameG L(table[fl[p]]), R(table[fr[p]]);
    const int tid = threadIdx.x, nR = R.n;
    uint32_t *s_start = reinterpret_cast<uint32_t *>(s_entry + nR);
    auto band = [&](int j, int &minr, int &maxr, int &o) {
        const float kpY = R.y[j];
        o = R.octave[j];
        const float r = __fmul_rn(2.0f, R.sf[o]);                       // :654
        maxr = (int)ceilf(__fadd_rn(kpY, r));                           // :655
        minr = (int)floorf(__fsub_rn(kpY, r));                          // :656
    };
    auto bin_of = [](int row) { return min(max(row, 0), ST_BINS - 1); };
    for (int b = tid; b < ST_BINS; b += ST_THREADS) s_cursor[b] = 0;
    __syncthreads();
    for (int j = tid; j < nR; j += ST_THREADS) {
        int minr, maxr, o;
        band(j, minr, maxr, o);
        atomicAdd(&s_cursor[bin_of(minr)], 1u);
    }
    __syncthreads();
    // exclusive prefix over the bins: each thread owns a contiguous run of bins
    {
        constexpr int PER = ST_BINS / ST_THREADS;
        __shared__ uint32_t s_part[ST_THREADS];
        uint32_t sum = 0;
        for (int k = 0; k < PER; k++) sum += s_cursor[tid * PER + k];
        s_part[tid] = sum;
        __syncthreads();
        uint32_t base = 0;
        for (int t = 0; t < tid; t++) base += s_part[t];
        for (int k = 0; k < PER; k++) {
            const uint32_t c = s_cursor[tid * PER + k];
            s_start[tid * PER + k] = base;
            s_cursor[tid * PER + k] = base;
            base += c;
        }
        if (tid == ST_THREADS - 1) s_start[ST_BINS] = base;
    }
    __syncthreads();
    for (int j = tid; j < nR; j += ST_THREADS) {
        int minr, maxr, o;
        band(j, minr, maxr, o);
        const uint32_t pos = atomicAdd(&s_cursor[bin_of(minr)], 1u);
        s_entry[pos] = make_uint4((uint32_t)j | ((uint32_t)o << 16), (uint32_t)minr, (uint32_t)maxr, __float_as_uint(R.x[j]));
    }
    __syncthreads();
    if (iL >= L.n) {                                                    // rows beyond the frame's keypoints: no match
        if (iL < stride) { best_idx[(size_t)p * stride + iL] = -1; best_dist[(size_t)p * stride + iL] = ORBM_TH_HIGH; }
        return;
    }
    const gptr<const v4u> g = (gptr<const v4u>)L.desc + 2 * (size_t)iL;
    const v4u qlo = g[0], qhi = g[1];
    const uint32_t q[8] = {qlo.x, qlo.y, qlo.z, qlo.w, qhi.x, qhi.y, qhi.z, qhi.w};
    const float uL = L.x[iL];
    const int row = (int)L.y[iL];                                       // vRowIndices[vL]: float -> index truncation (:684)
    const int levelL = L.octave[iL];
    const float minU = __fsub_rn(uL, max_d[p]);                         // :690
    uint32_t best = ((uint32_t)ORBM_TH_HIGH << 16) | 0xFFFFu;           // bestDist = TH_HIGH, no index (:696)
    if (!(uL < 0) && nR > 0) {                                          // maxU = uL - 0 (:691-694)
        float sf_max = 0.f;
        for (int lv = 0; lv < R.n_levels; lv++) sf_max = fmaxf(sf_max, R.sf[lv]);
        const int W = (int)ceilf(4.0f * sf_max) + 2;
        // bins are clamped, so the range is taken on clamped rows: a superset, every entry is tested exactly below
        const long long lo_row = (long long)row - W;
        const uint32_t lo = s_start[bin_of((int)max(lo_row, -1LL))], hi = s_start[bin_of(row) + 1];
        for (uint32_t e = lo; e < hi; e++) {
            const uint4 en = s_entry[e];
            const int o = (int)(en.x >> 16), j = (int)(en.x & 0xFFFFu);
            const float uR = __uint_as_float(en.w);
            if (row < (int)en.y || row > (int)en.z) continue;           // row band (:658-659)
            if (o < levelL - 1 || o > levelL + 1) continue;             // :712
            if (!(uR >= minU && uR <= uL)) continue;                    // :717
            const gptr<const v4u> c = (gptr<const v4u>)R.desc + 2 * (size_t)j;
            const v4u lo4 = c[0], hi4 = c[1];
            const uint32_t cd[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
            const uint32_t dist = hamming256(q, cd);
            if (dist < (uint32_t)ORBM_TH_HIGH) best = min(best, (dist << 16) | (uint32_t)j);   // :722-726
        }
    }
    const bool found = (best & 0xFFFFu) != 0xFFFFu || (best >> 16) < (uint32_t)ORBM_TH_HIGH;
    best_idx[(size_t)p * stride + iL] = found ? (int)(best & 0xFFFFu) : -1;
    best_dist[(size_t)p * stride + iL] = (int)(best >> 16);
}

int g_stereo_variant = 0;     // test knob: 1 = always the scan form

}  // namespace

extern "C" void orbm__stereo_variant(int v) { g_stereo_variant = v; }

extern "C" int orbm_stereo_match_batch(orbm_ctx *ctx, int n_pairs, const int32_t *frame_left, const int32_t *frame_right,
                                       const float *max_disparity, int32_t stride, int32_t *best_idx_r, int32_t *best_dist,
                                       int io_on_device) {
    if (!ctx) return ORBM_EINVAL;
    if (n_pairs < 0 || (n_pairs > 0 && (!frame_left || !frame_right || !max_disparity || !best_idx_r || !best_dist)) || stride < 0)
        return orbm_fail(ctx, ORBM_EINVAL, "stereo_match_batch: null argument or negative size");
    if (n_pairs == 0 || stride == 0) return ORBM_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!io_on_device) {
        int rc = orbm_check_frames(ctx, n_pairs, frame_left, "frame_left");
        if (!rc) rc = orbm_check_frames(ctx, n_pairs, frame_right, "frame_right");
        if (rc) return rc;
        for (int p = 0; p < n_pairs; p++)
            if (ctx->h_frames[frame_left[p]].n > stride)
                return orbm_fail(ctx, ORBM_EINVAL, "stereo_match_batch: stride %d < %d left keypoints of pair %d", stride, ctx->h_frames[frame_left[p]].n, p);
        rc = orbm_ws_reserve(ctx, (size_t)n_pairs * 12 + (size_t)n_pairs * stride * 8 + 4096);
        if (rc) return rc;
    }
    const size_t out_elems = (size_t)n_pairs * stride;
    IoStage io(ctx, io_on_device);
    const void *dl, *dr, *dm; void *di, *dd;
    int rc;
    if ((rc = io.in(frame_left, (size_t)n_pairs * 4, &dl)) || (rc = io.in(frame_right, (size_t)n_pairs * 4, &dr)) ||
        (rc = io.in(max_disparity, (size_t)n_pairs * 4, &dm)) || (rc = io.out(best_idx_r, out_elems * 4, &di)) ||
        (rc = io.out(best_dist, out_elems * 4, &dd)))
        return rc;
    // right frames that fit the LDS sort (host mode: known sizes; device mode: the caller's stride bounds the left side
    // only, so the scan form is used unless every live frame fits)
    int max_r = 0;
    if (!io_on_device) for (int p = 0; p < n_pairs; p++) max_r = std::max(max_r, ctx->h_frames[frame_right[p]].n);
    else for (size_t f = 0; f < ctx->h_frames.size(); f++) if (ctx->frame_arena[f] >= 0) max_r = std::max(max_r, ctx->h_frames[f].n);
    const bool small_batch = (long long)n_pairs * ((stride + 255) / 256) < 1024;
    if (max_r <= ST_SORT_MAX_N && g_stereo_variant != 1) {
        const size_t lds = (size_t)max_r * 16 + (ST_BINS + 1) * 4 + 16;
        if (small_batch) {
            const int blocks_per_pair = (stride + 63) / 64;
            if (lds > 48 * 1024) HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&stereo_sorted_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(stereo_sorted_kernel<64>, dim3((unsigned)(n_pairs * blocks_per_pair)), dim3(64), lds, ctx->stream, ctx->d_frames,
                               (const int32_t *)dl, (const int32_t *)dr, (const float *)dm, blocks_per_pair, stride, (int32_t *)di, (int32_t *)dd);
        } else {
            const int blocks_per_pair = (stride + 255) / 256;
            if (lds > 48 * 1024) HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&stereo_sorted_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(stereo_sorted_kernel<256>, dim3((unsigned)(n_pairs * blocks_per_pair)), dim3(256), lds, ctx->stream, ctx->d_frames,
                               (const int32_t *)dl, (const int32_t *)dr, (const float *)dm, blocks_per_pair, stride, (int32_t *)di, (int32_t *)dd);
        }
    } else if (small_batch) {
        const int blocks_per_pair = (stride + 63) / 64;
        hipLaunchKernelGGL(stereo_kernel<64>, dim3((unsigned)(n_pairs * blocks_per_pair)), dim3(64), 0, ctx->stream, ctx->d_frames,
                           (const int32_t *)dl, (const int32_t *)dr, (const float *)dm, blocks_per_pair, stride, (int32_t *)di, (int32_t *)dd);
    } else {
        const int blocks_per_pair = (stride + 255) / 256;
        hipLaunchKernelGGL(stereo_kernel<256>, dim3((unsigned)(n_pairs * blocks_per_pair)), dim3(256), 0, ctx->stream, ctx->d_frames,
                           (const int32_t *)dl, (const int32_t *)dr, (const float *)dm, blocks_per_pair, stride, (int32_t *)di, (int32_t *)dd);
    }
    HIPCHK(ctx, hipGetLastError());
    return io.finish();
}
