// bf2nn.hip -- brute-force 2-NN + threshold + ratio test over keyframe pairs (BASELINE config 2 and the
// brute-force variant of config 5).  Integer-VALU (popcount) bound: 8 v_xor_b32 + 8 v_bcnt_u32_b32 per
// 256-bit comparison plus 3 tracking ops (v_lshl_or_b32 key, v_med3_u32 second, v_min_u32 best).
//
// Work decomposition: one 256-thread workgroup owns 256*R consecutive query rows of frame a (R rows per
// lane, 8 VGPRs each) and walks ALL candidate rows of frame b in ascending index.  Candidate rows are
// staged global -> LDS in coalesced 16-byte loads, TILE rows at a time; every lane then reads the same
// LDS row (broadcast ds_read_b128 x2) and reuses it for its R queries.  Since each lane sees the
// candidates in the reference's visit order, key = (dist << 16) | j reproduces the strict-< first-wins
// rule of ORBmatcher.cc:290-300, and second-best is the median of (best, second, key).
#include <string.h>
#include <algorithm>
#include "orbm_internal.h"

namespace {

constexpr int BF_TILE = 512;  // candidate rows per LDS tile (16 KB)
constexpr int BF_UNROLL = 4;

template <int R>
__global__ __launch_bounds__(256) void bf2nn_kernel(const FrameDev *__restrict__ table,
                                                    const int32_t *__restrict__ fa, const int32_t *__restrict__ fb,
                                                    int blocks_per_pair, float ratio, int th, int stride,
                                                    int32_t *__restrict__ match, int32_t *__restrict__ best_d,
                                                    int32_t *__restrict__ second_d, int32_t *__restrict__ counts) {
    __shared__ uint4 tile[BF_TILE * 2];
    const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
    const int p = bid / blocks_per_pair;
    const int qb = bid - p * blocks_per_pair;
    const FrameDev &A = table[fa[p]];
    const FrameDev &B = table[fb[p]];
    const int nA = A.n, nB = B.n;
    const int tid = threadIdx.x;
    const int q0 = qb * (256 * R) + tid;
    if (qb * (256 * R) >= nA) return;  // whole block idle (uniform)

    uint32_t q[R][8];
    uint32_t best[R], second[R];
    const uint4 *Ad = reinterpret_cast<const uint4 *>(A.desc);
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int qi = q0 + r * 256;
        uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
        if (qi < nA) { lo = Ad[2 * qi]; hi = Ad[2 * qi + 1]; }
        q[r][0] = lo.x; q[r][1] = lo.y; q[r][2] = lo.z; q[r][3] = lo.w;
        q[r][4] = hi.x; q[r][5] = hi.y; q[r][6] = hi.z; q[r][7] = hi.w;
        best[r] = (256u << 16) | 0xFFFFu;   // bestDist1 = 256, bestIdx = -1  (ORBmatcher.cc:271-273)
        second[r] = (256u << 16) | 0xFFFFu; // bestDist2 = 256
    }

    const uint4 *Bd = reinterpret_cast<const uint4 *>(B.desc);
    for (int t0 = 0; t0 < nB; t0 += BF_TILE) {
        const int rows = min(BF_TILE, nB - t0);
        __syncthreads();
        for (int i = tid; i < rows * 2; i += 256) tile[i] = Bd[t0 * 2 + i];
        __syncthreads();
        // inline asm is convergent in HIP, so a runtime-trip-count loop is not unrolled by hipcc: unroll
        // by hand (constant inner trip count) and finish the tail one row at a time.
        auto step = [&](int j) {
            const uint4 c0 = tile[2 * j], c1 = tile[2 * j + 1];
            const uint32_t c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            const uint32_t jj = (uint32_t)(t0 + j);
#pragma unroll
            for (int r = 0; r < R; r++) {
                const uint32_t d = hamming256(q[r], c);
                const uint32_t key = (d << 16) | jj;
                second[r] = umed3(best[r], second[r], key);
                best[r] = min(best[r], key);
            }
        };
        int j = 0;
        for (; j + BF_UNROLL <= rows; j += BF_UNROLL) {
#pragma unroll
            for (int u = 0; u < BF_UNROLL; u++) step(j + u);
        }
        for (; j < rows; j++) step(j);
    }

    int accepted = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int qi = q0 + r * 256;
        if (qi < nA) {
            const int bd = (int)(best[r] >> 16), sd = (int)(second[r] >> 16);
            const int bidx = bd >= 256 ? -1 : (int)(best[r] & 0xFFFFu);
            int m = -1;
            if (bd <= th && (float)bd < __fmul_rn(ratio, (float)sd)) { m = bidx; accepted++; }  // :304-308
            const size_t o = (size_t)p * stride + qi;
            match[o] = m;
            if (best_d) best_d[o] = bd;
            if (second_d) second_d[o] = sd;
        }
    }
    // one atomic per wave
    for (int off = 32; off > 0; off >>= 1) accepted += __shfl_down(accepted, off, 64);
    if ((tid & 63) == 0 && accepted) atomicAdd(&counts[p], accepted);
}

}  // namespace

extern "C" int orbm_bf2nn_batch(orbm_ctx *ctx, int n_pairs, const int32_t *frame_a, const int32_t *frame_b,
                                float ratio, int th, int32_t stride, int32_t *match, int32_t *best_d,
                                int32_t *second_d, int32_t *counts, int io_on_device) {
    if (!ctx) return ORBM_EINVAL;
    if (n_pairs < 0 || (n_pairs > 0 && (!frame_a || !frame_b || !match || !counts)) || stride < 0)
        return orbm_fail(ctx, ORBM_EINVAL, "bf2nn_batch: null argument or negative size");
    if (n_pairs == 0) return ORBM_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int max_na = 0;
    if (!io_on_device) {
        int rc = orbm_check_frames(ctx, n_pairs, frame_a, "frame_a");
        if (!rc) rc = orbm_check_frames(ctx, n_pairs, frame_b, "frame_b");
        if (rc) return rc;
        for (int p = 0; p < n_pairs; p++) max_na = std::max(max_na, ctx->h_frames[frame_a[p]].n);
        if (max_na > stride) return orbm_fail(ctx, ORBM_EINVAL, "bf2nn_batch: stride %d < largest a-frame %d", stride, max_na);
    } else {
        // device-resident pair lists cannot be inspected here: the stride bounds the rows any block touches,
        // and ids are the caller's contract (orbm_frames_upload handles, all live).
        max_na = stride;
    }
    if (max_na == 0) {
        if (!io_on_device) memset(counts, 0, (size_t)n_pairs * 4);
        else HIPCHK(ctx, hipMemsetAsync(counts, 0, (size_t)n_pairs * 4, ctx->stream));
        return ORBM_OK;
    }
    const size_t out_elems = (size_t)n_pairs * stride;
    IoStage io(ctx, io_on_device);
    if (!io_on_device) {
        size_t need = (size_t)n_pairs * 12 + out_elems * 4 * (1 + (best_d != nullptr) + (second_d != nullptr)) + 4096;
        int rc = orbm_ws_reserve(ctx, need);
        if (rc) return rc;
    }
    const void *dfa, *dfb; void *dm, *dbd, *dsd, *dc;
    int rc;
    if ((rc = io.in(frame_a, (size_t)n_pairs * 4, &dfa)) || (rc = io.in(frame_b, (size_t)n_pairs * 4, &dfb)) ||
        (rc = io.out(match, out_elems * 4, &dm)) || (rc = io.out(best_d, out_elems * 4, &dbd)) ||
        (rc = io.out(second_d, out_elems * 4, &dsd)) || (rc = io.out(counts, (size_t)n_pairs * 4, &dc)))
        return rc;
    HIPCHK(ctx, hipMemsetAsync(dc, 0, (size_t)n_pairs * 4, ctx->stream));
    if (!io_on_device) HIPCHK(ctx, hipMemsetAsync(dm, 0xFF, out_elems * 4, ctx->stream));  // rows beyond n stay -1
    constexpr int R = 2;
    const int bpp = (max_na + 256 * R - 1) / (256 * R);
    const long long grid = (long long)n_pairs * bpp;
    if (grid > 0x7FFFFFFFLL) return orbm_fail(ctx, ORBM_ELIMIT, "bf2nn_batch: grid too large");
    hipLaunchKernelGGL(bf2nn_kernel<R>, dim3((unsigned)grid), dim3(256), 0, ctx->stream, ctx->d_frames,
                       (const int32_t *)dfa, (const int32_t *)dfb, bpp, ratio, th, stride, (int32_t *)dm,
                       (int32_t *)dbd, (int32_t *)dsd, (int32_t *)dc);
    HIPCHK(ctx, hipGetLastError());
    return io.finish();
}
