// bf2nn.hip -- brute-force 2-NN + threshold + ratio test over keyframe pairs (BASELINE config 2 and the
// brute-force variant of config 5).  Three forms with identical results:
//   * bf2nn_mfma4_kernel (default): Hamming distances as exact +-1 dot products on the matrix cores (MX-scaled FP4 MFMA);
//   * bf2nn_mfma_kernel: the same through int8 MFMA;
//   * bf2nn_kernel: xor + popcount on the vector ALU (described next), the form the reference's arithmetic maps to
//     directly; kept as the cross-check of the other two and for callers that do not want the +-1 frame images.
//
// Measured on gfx950 (scripts/ubench_valu2.hip): v_xor/v_and/v_or/v_add issue in ~2 cycles per wave64
// instruction, v_bcnt_u32_b32 / v_min_u32 / v_med3_u32 / v_lshl_or_b32 / v_cmp in ~4.  One 256-bit
// comparison needs 8 xor + 8 bcnt no matter what; everything else is overhead, so the inner loop keeps
// the per-comparison extras down to ONE fast op:
//   * each lane owns R query rows in VGPRs and walks ALL candidate rows in ascending index, so the
//     reference's visit order (strict-< first-wins, ORBmatcher.cc:290-300) is the loop order;
//   * the popcount chain starts from bias = 0x80000000 - second_best instead of 0, so bit 31 of the
//     result is CLEAR exactly when dist < second_best, i.e. when the reference's update would fire;
//   * a group of U candidates x R queries is AND-ed into one mask; only if some lane of the wave has a
//     clear bit 31 does the wave take the (rare) exact update: key = (dist << 16) | j,
//     second = med3(best, second, key), best = min(best, key);
//   * when the caller does not ask for the raw distances, candidates at distance >= track_below can never
//     change the outcome of "best <= th && (float)best < ratio*(float)second" (track_below is the smallest
//     T > th with (float)th < ratio*(float)T), so the bias is capped at track_below and the exact update
//     only fires for candidates that can matter.  All nA x nB distances are still evaluated.
// Candidate rows are staged global -> LDS in coalesced 16-byte loads, TILE rows at a time; all lanes read
// the same LDS row (broadcast ds_read_b128 x2) and reuse it for their R queries.
#include <type_traits>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "orbm_internal.h"

namespace {

constexpr int BF_TILE = 512;  // candidate rows per LDS tile (16 KB)
constexpr int BF_R = 2;       // query rows per lane   (R2/U4: 47 VGPRs, 8 waves per SIMD; fastest of the
constexpr int BF_U = 4;       // candidate rows per group  variants A/B-ed in scripts/bf_variants.py)

// popcount chain seeded with `bias`: returns bias + hamming(a, b)
__device__ __forceinline__ uint32_t hamming256_biased(const uint32_t (&a)[8], const uint32_t (&b)[8], uint32_t bias) {
    uint32_t d;
    {
        const uint32_t x = a[0] ^ b[0];
        asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(bias));
    }
#pragma unroll
    for (int i = 1; i < 8; i++) {
        const uint32_t x = a[i] ^ b[i];
        asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d) : "v"(x));
    }
    return d;
}

// the same chain in two pieces (early-abandon variant): words [0, 6) seeded with the bias, then words [6, 8) on top
__device__ __forceinline__ uint32_t hamming192_biased(const uint32_t (&a)[8], const uint32_t (&b)[6], uint32_t bias) {
    uint32_t d;
    {
        const uint32_t x = a[0] ^ b[0];
        asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(bias));
    }
#pragma unroll
    for (int i = 1; i < 6; i++) {
        const uint32_t x = a[i] ^ b[i];
        asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d) : "v"(x));
    }
    return d;
}
__device__ __forceinline__ uint32_t hamming64_more(const uint32_t (&a)[8], uint32_t b6, uint32_t b7, uint32_t d) {
    const uint32_t x6 = a[6] ^ b6, x7 = a[7] ^ b7;
    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d) : "v"(x6));
    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d) : "v"(x7));
    return d;
}

// DIAG 0: product kernel.  1 / 2: timing diagnostics.  3: EARLY ABANDON (opt-in, reported separately by bench.py): a
// distance only grows, so when the first 192 bits of every comparison of a group already reach the tracking bound the
// last 64 bits cannot change anything and are skipped (same results; not every distance is evaluated in full).
template <int R, int U, int DIAG>
__global__ __launch_bounds__(256) void bf2nn_kernel(const FrameDev *__restrict__ table,
                                                    const int32_t *__restrict__ fa, const int32_t *__restrict__ fb,
                                                    int blocks_per_pair, float ratio, int th, int track_below, int stride,
                                                    int32_t *__restrict__ match, int32_t *__restrict__ best_d,
                                                    int32_t *__restrict__ second_d, int32_t *__restrict__ counts) {
    __shared__ uint4 tile[BF_TILE * 2];
    const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
    const int p = bid / blocks_per_pair;
    const int qb = bid - p * blocks_per_pair;
    const FrameG A(table[fa[p]]), B(table[fb[p]]);
    const int nA = A.n, nB = B.n;
    const int tid = threadIdx.x;
    const int q0 = qb * (256 * R) + tid;
    if (qb * (256 * R) >= nA) {        // whole block idle (uniform): its rows read "no match" up to the stride
        for (int r = 0; r < R; r++) if (q0 + r * 256 < stride) match[(size_t)p * stride + q0 + r * 256] = -1;
        return;
    }

    uint32_t q[R][8];
    uint32_t best[R], second[R], bias[R];
    const gptr<const v4u> Ad = (gptr<const v4u>)A.desc;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int qi = q0 + r * 256;
        v4u lo = {0, 0, 0, 0}, hi = lo;
        if (qi < nA) { lo = Ad[2 * qi]; hi = Ad[2 * qi + 1]; }
        q[r][0] = lo.x; q[r][1] = lo.y; q[r][2] = lo.z; q[r][3] = lo.w;
        q[r][4] = hi.x; q[r][5] = hi.y; q[r][6] = hi.z; q[r][7] = hi.w;
        best[r] = (256u << 16) | 0xFFFFu;   // bestDist1 = 256, bestIdx = -1  (ORBmatcher.cc:271-273)
        second[r] = (256u << 16) | 0xFFFFu; // bestDist2 = 256
        bias[r] = 0x80000000u - (uint32_t)min(256, track_below);
    }

    const gptr<const v4u> Bd = (gptr<const v4u>)B.desc;
    for (int t0 = 0; t0 < nB; t0 += BF_TILE) {
        const int rows = min(BF_TILE, nB - t0);
        __syncthreads();
        for (int i = tid; i < rows * 2; i += 256) {
            const v4u row = Bd[t0 * 2 + i];
            tile[i] = make_uint4(row.x, row.y, row.z, row.w);
        }
        __syncthreads();
        int j = 0;
        for (; j + U <= rows; j += U) {
            uint32_t acc[U][R];
            uint32_t m = 0xFFFFFFFFu;
            if (DIAG == 3) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint4 c0 = tile[2 * (j + u)];
                    const uint2 c1 = *reinterpret_cast<const uint2 *>(&tile[2 * (j + u) + 1]);
                    const uint32_t c[6] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y};
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        acc[u][r] = hamming192_biased(q[r], c, bias[r]);
                        m &= acc[u][r];
                    }
                }
                if (!__any((int)m >= 0)) continue;       // every partial distance already at the bound: group done
                m = 0xFFFFFFFFu;
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint2 c2 = *(reinterpret_cast<const uint2 *>(&tile[2 * (j + u) + 1]) + 1);
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        acc[u][r] = hamming64_more(q[r], c2.x, c2.y, acc[u][r]);
                        m &= acc[u][r];
                    }
                }
            } else
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint4 c0 = tile[2 * (j + u)], c1 = tile[2 * (j + u) + 1];
                const uint32_t c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
                for (int r = 0; r < R; r++) {
                    acc[u][r] = hamming256_biased(q[r], c, bias[r]);
                    m &= acc[u][r];
                }
            }
            // DIAG 1: never update (timing floor, wrong results); DIAG 2: always update (v1 behaviour)
            if (DIAG != 1 && (DIAG == 2 || __any((int)m >= 0))) {  // some lane saw dist < second best: exact update for the whole group
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint32_t jj = (uint32_t)(t0 + j + u);
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const uint32_t key = ((acc[u][r] - bias[r]) << 16) | jj;
                        second[r] = umed3(best[r], second[r], key);
                        best[r] = min(best[r], key);
                    }
                }
#pragma unroll
                for (int r = 0; r < R; r++) bias[r] = 0x80000000u - min(second[r] >> 16, (uint32_t)track_below);
            }
        }
        for (; j < rows; j++) {  // tail of the tile
            const uint4 c0 = tile[2 * j], c1 = tile[2 * j + 1];
            const uint32_t c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            const uint32_t jj = (uint32_t)(t0 + j);
#pragma unroll
            for (int r = 0; r < R; r++) {
                const uint32_t key = (hamming256(q[r], c) << 16) | jj;
                second[r] = umed3(best[r], second[r], key);
                best[r] = min(best[r], key);
                bias[r] = 0x80000000u - min(second[r] >> 16, (uint32_t)track_below);
            }
        }
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
        } else if (qi < stride) {
            match[(size_t)p * stride + qi] = -1;        // rows between the frame's n and the stride: no match
        }
    }
    // one atomic per wave
    for (int off = 32; off > 0; off >>= 1) accepted += __shfl_down(accepted, off, 64);
    if ((tid & 63) == 0 && accepted) atomicAdd(&counts[p], accepted);
}

// ---- MFMA form ---------------------------------------------------------------------------------------------------------
// With every bit as a +-1 int8 (FrameDev::desc8), the dot product of two rows is (#equal bits - #different bits) =
// 256 - 2 * hamming: one v_mfma_i32_32x32x32_i8 produces 32 x 32 such dots over 32 of the 256 bit positions, 8 of them a
// tile of 1024 exact Hamming distances in 8 x 32 cycles -- against ~960 cycles of xor + popcount on the vector ALU.
// Operand A = 32 candidate rows (from LDS), operand B = 32 query rows (held in registers for the whole kernel), so in the
// result a lane holds ONE query (column lane & 31) and 16 candidates (rows (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)).
// Any k order works as long as A and B use the same one (both fragments are bytes [32 s + 16 h, + 16) of a row).
// Per tile and query the 16 dots are folded with v_max3_i32 and compared with the lane's threshold
// T = 256 - 2 * min(second best, track_below): only if some lane of the wave beats it does the wave run the exact update
// (key = dist << 16 | index; second = med3(best, second, key); best = min(best, key)) -- the same rule and the same
// track_below argument as bf2nn_kernel above.  The two lanes that share a query merge their top-2 at the end.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int MF_ROWS = 128;       // candidate rows per LDS stage (32 KB of int8)

template <bool WANT_DIST, int DIAG>
__global__ __launch_bounds__(256) void bf2nn_mfma_kernel(const FrameDev *__restrict__ table,
                                                         const int32_t *__restrict__ fa, const int32_t *__restrict__ fb,
                                                         int blocks_per_pair, float ratio, int th, int track_below, int stride,
                                                         int32_t *__restrict__ match, int32_t *__restrict__ best_d,
                                                         int32_t *__restrict__ second_d, int32_t *__restrict__ counts) {
    // 16-byte chunk c of row r sits at chunk position c ^ (r & 15): conflict-free ds_read_b128 for 16 rows at one k step
    __shared__ v4i tile[MF_ROWS * 16];
    const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
    const int p = bid / blocks_per_pair;
    const int qb = bid - p * blocks_per_pair;
    const FrameG A(table[fa[p]]), B(table[fb[p]]);
    const int nA = A.n, nB = B.n;
    if (qb * 256 >= nA) {              // whole block idle (uniform): its rows read "no match" up to the stride
        if (qb * 256 + (int)threadIdx.x < stride) match[(size_t)p * stride + qb * 256 + threadIdx.x] = -1;
        return;
    }
    const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63, col = l & 31, h = l >> 5;
    const int nA8 = max(32, (nA + 31) & ~31), nB8 = (nB + 31) & ~31;
    const int qbase = qb * 256 + wave * 64;

    v4i bq[2][8];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int row = min(qbase + t * 32 + col, nA8 - 1);
        const gptr<const v4i> src = (gptr<const v4i>)A.desc8 + (size_t)row * 16;
#pragma unroll
        for (int s = 0; s < 8; s++) bq[t][s] = src[2 * s + h];
    }
    uint32_t best[2], second[2];
    int T[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        best[t] = (256u << 16) | 0xFFFFu;
        second[t] = (256u << 16) | 0xFFFFu;
        T[t] = 256 - 2 * min(256, track_below);
    }

    const gptr<const v4i> Bd8 = (gptr<const v4i>)B.desc8;
    for (int c0 = 0; c0 < nB8; c0 += MF_ROWS) {
        const int rows = min(MF_ROWS, nB8 - c0);            // a multiple of 32
        if (DIAG != 2 || c0 == 0) {
            // all loads of the stage first, then the LDS writes (a load-wait-write loop costs one L2 round trip per row group)
            v4i stg[MF_ROWS * 16 / 256];
#pragma unroll
            for (int k = 0; k < MF_ROWS * 16 / 256; k++) {
                const int i = min(tid + 256 * k, rows * 16 - 1);          // unconditional (clamped): the array stays in registers
                stg[k] = Bd8[(size_t)c0 * 16 + i];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < MF_ROWS * 16 / 256; k++) {
                const int i = tid + 256 * k, r = i >> 4, c = i & 15;
                if (i < rows * 16) tile[r * 16 + (c ^ (r & 15))] = stg[k];
            }
            __syncthreads();
        }
        for (int rb = 0; rb < rows; rb += 32) {
            v16i acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc0;
            const v4i *arow = tile + (rb + col) * 16;
#pragma unroll
            for (int s = 0; s < 8; s++) {
                const v4i a = arow[(2 * s + h) ^ (col & 15)];
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[0][s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[1][s], acc1, 0, 0, 0);
            }
            int m0 = max(max(acc0[0], acc0[1]), acc0[2]), m1 = max(max(acc1[0], acc1[1]), acc1[2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) {
                m0 = max(max(m0, acc0[r]), acc0[r + 1]);
                m1 = max(max(m1, acc1[r]), acc1[r + 1]);
            }
            m0 = max(m0, acc0[15]); m1 = max(m1, acc1[15]);
            if (DIAG == 1) { best[0] += (uint32_t)(m0 + m1); continue; }
            if (__any(m0 > T[0] || m1 > T[1])) {            // some lane saw dist < its bound: exact update of the tile
                const int cand0 = c0 + rb + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int cand = cand0 + (r & 3) + 8 * (r >> 2);
                    const uint32_t k0 = cand < nB ? ((uint32_t)((256 - acc0[r]) >> 1) << 16) | (uint32_t)cand : 0xFFFFFFFFu;
                    const uint32_t k1 = cand < nB ? ((uint32_t)((256 - acc1[r]) >> 1) << 16) | (uint32_t)cand : 0xFFFFFFFFu;
                    second[0] = umed3(best[0], second[0], k0); best[0] = min(best[0], k0);
                    second[1] = umed3(best[1], second[1], k1); best[1] = min(best[1], k1);
                }
#pragma unroll
                for (int t = 0; t < 2; t++) T[t] = 256 - 2 * (int)min(second[t] >> 16, (uint32_t)track_below);
            }
        }
    }

    int accepted = 0;
#pragma unroll
    for (int t = 0; t < 2; t++) {
        // the two lanes of a query hold the top-2 of disjoint candidate sets: the two smallest keys of the four
        const uint32_t pb = (uint32_t)__shfl_xor((int)best[t], 32), ps = (uint32_t)__shfl_xor((int)second[t], 32);
        const uint32_t b = min(best[t], pb), s2 = min(max(best[t], pb), min(second[t], ps));
        const int qi = qbase + t * 32 + col;
        if (h == 0 && qi < nA) {
            const int bd = (int)(b >> 16), sd = (int)(s2 >> 16);
            const int bidx = bd >= 256 ? -1 : (int)(b & 0xFFFFu);
            int mm = -1;
            if (bd <= th && (float)bd < __fmul_rn(ratio, (float)sd)) { mm = bidx; accepted++; }  // :304-308
            const size_t o = (size_t)p * stride + qi;
            match[o] = mm;
            if (WANT_DIST) {
                if (best_d) best_d[o] = bd;
                if (second_d) second_d[o] = sd;
            }
        } else if (h == 0 && qi < stride) {
            match[(size_t)p * stride + qi] = -1;        // rows between the frame's n and the stride: no match
        }
    }
    for (int off = 32; off > 0; off >>= 1) accepted += __shfl_down(accepted, off, 64);
    if (l == 0 && accepted) atomicAdd(&counts[p], accepted);
}

// ---- FP4 form ----------------------------------------------------------------------------------------------------------
// The same product with the +-1 values as FP4 E2M1 (FrameDev::desc4, 128 B per row) through
// v_mfma_f32_32x32x64_f8f6f4 (the builtin with zero scale operands selects the unscaled 8-byte encoding, block scale 2^0;
// the scaled encoding with explicit 2^0 scales is a second instruction word and measured 2.5 % slower): K = 64 per instruction at the cycles of the bf16 32x32 form, so a
// tile of 1024 distances is 4 instructions instead of 8, and every staged byte carries two bit positions.  (The 16x16x128
// shape of the same instruction was measured 12 % slower here, 8-wave workgroups and register-prefetched stages no
// faster; scripts/ubench_mfma4*.hip: what the chip sustains on this instruction and what each ingredient of this loop costs.)  Products are
// +-1 and the f32 sums are integers of magnitude <= 256: exact.  Lane l supplies bytes [32 s + 16 h, + 16) of row l & 31
// for k step s (32 values), for A and B alike.  C/D layout and everything after the product as in the int8 form.
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int MF4_ROWS = 256;      // candidate rows in LDS (32 KB of FP4), two buffers of half of it
constexpr int MF4_NT = 2;          // query tiles (of 32) per wave (4 was measured no faster and wastes more on small frames)

template <bool WANT_DIST, int DIAG, int NT>
__global__ __launch_bounds__(256) void bf2nn_mfma4_kernel(const FrameDev *__restrict__ table,
                                                          const int32_t *__restrict__ fa, const int32_t *__restrict__ fb,
                                                          int blocks_per_pair, float ratio, int th, int track_below, int stride,
                                                          int32_t *__restrict__ match, int32_t *__restrict__ best_d,
                                                          int32_t *__restrict__ second_d, int32_t *__restrict__ counts) {
    // 16-byte chunk c (of 8) of row r sits at chunk position c ^ ((r >> 1) & 7): two rows share a 256-byte bank span
    __shared__ v4i tile[MF4_ROWS * 8];
    const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
    const int p = bid / blocks_per_pair;
    const int qb = bid - p * blocks_per_pair;
    const FrameG A(table[fa[p]]), B(table[fb[p]]);
    const int nA = A.n, nB = B.n;
    constexpr int QW = 32 * NT;        // queries per wave: NT tiles of 32
    if (qb * (4 * QW) >= nA) {         // whole block idle (uniform): its rows read "no match" up to the stride
        for (int i = threadIdx.x; i < 4 * QW; i += 256) if (qb * (4 * QW) + i < stride) match[(size_t)p * stride + qb * (4 * QW) + i] = -1;
        return;
    }
    const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63, col = l & 31, h = l >> 5;
    const int nA8 = max(32, (nA + 31) & ~31), nB8 = (nB + 31) & ~31;
    const int qbase = qb * (4 * QW) + wave * QW;

    v8i bq[NT][4];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int row = min(qbase + t * 32 + col, nA8 - 1);
        const gptr<const v4i> src = (gptr<const v4i>)A.desc4 + (size_t)row * 8;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const v4i x = src[2 * s + h];
            bq[t][s] = (v8i){x[0], x[1], x[2], x[3], 0, 0, 0, 0};
        }
    }
    uint32_t best[NT], second[NT];
    float T[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        best[t] = (256u << 16) | 0xFFFFu;
        second[t] = (256u << 16) | 0xFFFFu;
        T[t] = (float)(256 - 2 * min(256, track_below));
    }

    const gptr<const v4i> Bd4 = (gptr<const v4i>)B.desc4;
    // Two LDS buffers of MF4_ROWS / 2 rows; the next stage travels global -> LDS directly (global_load_lds, no registers)
    // while the current one is multiplied; one barrier per stage.  A wave instruction fills 1 KB = 8 rows of LDS in lane
    // order, so the swizzle is applied to the global address each lane reads: LDS chunk position (r, cpos) holds chunk
    // cpos ^ ((r >> 1) & 7) of row r.  Rows past the end of the frame image are clamped (never multiplied).
    constexpr int HALF = MF4_ROWS / 2;                      // rows per buffer
    constexpr int INSTR_PER_WAVE = HALF * 8 / 256;          // 16-byte chunks per buffer / (4 waves x 64 lanes)
    auto issue_stage = [&](int c0, int buf) {
#pragma unroll
        for (int j = 0; j < INSTR_PER_WAVE; j++) {
            const int g0 = __builtin_amdgcn_readfirstlane((wave * INSTR_PER_WAVE + j) * 64);
            const int g = g0 + l, r = g >> 3, cpos = g & 7;
            const int row = min(c0 + r, nB8 - 1);
            const gptr<const v4i> src = Bd4 + (size_t)row * 8 + (cpos ^ ((r >> 1) & 7));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(tile + buf * (HALF * 8) + g0), 16, 0, 0);
        }
    };
    if (nB8 > 0) issue_stage(0, 0);
    int stage = 0;
    for (int c0 = 0; c0 < nB8; c0 += HALF, stage ^= 1) {
        const int rows = min(HALF, nB8 - c0);               // a multiple of 32
        // this wave's part of stage `stage` has landed: LDS-DMA loads count on vmcnt, which the barrier's own fence does
        // not wait for (it covers LDS instructions only); they were issued a whole stage ago, so the wait costs nothing
        __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0)
        __syncthreads();                                    // every wave's part has: the stage is complete, the other buffer is free
        if (DIAG != 2 && c0 + HALF < nB8) issue_stage(c0 + HALF, stage ^ 1);
        const v4i *tbuf = tile + (DIAG == 2 ? 0 : stage) * (HALF * 8);
        for (int rb = 0; rb < rows; rb += 32) {
            v16f acc[NT];
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = (v16f){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            const v4i *arow = tbuf + (rb + col) * 8;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const v4i x = arow[(2 * s + h) ^ ((col >> 1) & 7)];
                const v8i a = {x[0], x[1], x[2], x[3], 0, 0, 0, 0};
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bq[t][s], acc[t], 4, 4, 0, 0, 0, 0);
            }
            bool fire = false;
            float msum = 0.f;
#pragma unroll
            for (int t = 0; t < NT; t++) {
                float m = fmaxf(fmaxf(acc[t][0], acc[t][1]), acc[t][2]);
#pragma unroll
                for (int r = 3; r < 15; r += 2) m = fmaxf(fmaxf(m, acc[t][r]), acc[t][r + 1]);
                m = fmaxf(m, acc[t][15]);
                fire = fire || m > T[t];
                msum += m;
            }
            if (DIAG == 1) { best[0] += (uint32_t)(int)msum; continue; }
            if (__any(fire)) {                              // some lane saw dist < its bound: exact update of the step
                const int cand0 = c0 + rb + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int cand = cand0 + (r & 3) + 8 * (r >> 2);
#pragma unroll
                    for (int t = 0; t < NT; t++) {
                        const uint32_t key = cand < nB ? ((uint32_t)((256 - (int)acc[t][r]) >> 1) << 16) | (uint32_t)cand : 0xFFFFFFFFu;
                        second[t] = umed3(best[t], second[t], key); best[t] = min(best[t], key);
                    }
                }
#pragma unroll
                for (int t = 0; t < NT; t++) T[t] = (float)(256 - 2 * (int)min(second[t] >> 16, (uint32_t)track_below));
            }
        }
    }

    int accepted = 0;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const uint32_t pb = (uint32_t)__shfl_xor((int)best[t], 32), ps = (uint32_t)__shfl_xor((int)second[t], 32);
        const uint32_t b = min(best[t], pb), s2 = min(max(best[t], pb), min(second[t], ps));
        const int qi = qbase + t * 32 + col;
        if (h == 0 && qi < nA) {
            const int bd = (int)(b >> 16), sd = (int)(s2 >> 16);
            const int bidx = bd >= 256 ? -1 : (int)(b & 0xFFFFu);
            int mm = -1;
            if (bd <= th && (float)bd < __fmul_rn(ratio, (float)sd)) { mm = bidx; accepted++; }  // :304-308
            const size_t o = (size_t)p * stride + qi;
            match[o] = mm;
            if (WANT_DIST) {
                if (best_d) best_d[o] = bd;
                if (second_d) second_d[o] = sd;
            }
        } else if (h == 0 && qi < stride) {
            match[(size_t)p * stride + qi] = -1;        // rows between the frame's n and the stride: no match
        }
    }
    for (int off = 32; off > 0; off >>= 1) accepted += __shfl_down(accepted, off, 64);
    if (l == 0 && accepted) atomicAdd(&counts[p], accepted);
}


// ---- the FP4 form again, WITHOUT LDS (variant 12) ----------------------------------------------------------------------
// Candidate rows come straight from the frame's raw descriptors (32 B per row: lane (col, h) reads the 16-byte half h of row
// col, one coalesced 1 KB read per 32 rows) and become FP4 +-1 values in registers (3 shifts + 4 v_and_or_b32 per 32 bits,
// as in bowm.hip; the order of the 256 bit positions inside the product is free as long as both operands use the same one).
// No staging instructions, no barrier, a quarter of the operand bytes; 28 more vector instructions per 32 candidate rows.
// The shift is an asm statement (hipcc would otherwise turn (w << k) & sign into (w & m) << k and spend an instruction more);
// the and-or is C ON PURPOSE: it is the instruction that writes a matrix operand, and a matrix instruction needs two wait
// states after a vector write of one of its operands, which hipcc inserts only behind instructions it knows -- an asm
// statement is not one (found in bowm.hip: with one wait state the first passes of the product read the OLD register).
template <int S>
__device__ __forceinline__ uint32_t bf_shl(uint32_t x) {
    uint32_t r;
    asm("v_lshlrev_b32 %0, %1, %2" : "=v"(r) : "n"(S), "v"(x));
    return r;
}
__device__ __forceinline__ uint32_t bf_and_or(uint32_t x, uint32_t bits) { return (x & 0x88888888u) | bits; }
__device__ __forceinline__ v8i bf_expand_fp4(uint32_t w, uint32_t ones) {
    return (v8i){(int)bf_and_or(bf_shl<3>(w), ones), (int)bf_and_or(bf_shl<2>(w), ones), (int)bf_and_or(bf_shl<1>(w), ones),
                 (int)bf_and_or(w, ones), 0, 0, 0, 0};
}

// Operands without a sign (PIPE 2).  The chip holds 2.0-2.1 GHz on +-1 operands and 2.25-2.3 GHz on non-negative ones
// (scripts/ubench_mfma4c.hip: zeros 0.98 of the nominal peak, +-1 x +-1 0.82, {0,1} x +-1 0.81, {0,1} x {0,1} 0.93 -- the
// sign is what costs the power, not the density).  A query bit becomes 0 / weight, a candidate bit the COMPLEMENT as
// 0 / (2 / weight): the product sums to 2 * #(q = 1, c = 0) = 2 * (popcount(q) - popcount(q & c)), and with the candidate
// row's popcount as the accumulator's initial value a result is  popcount(c) + 2 popcount(q) - 2 popcount(q & c)
// = d + popcount(q):  the distance plus a per-lane constant, every value an integer in [0, 768], exact in f32.
// Dword k of an operand takes bits k, k + 4, ... of the raw dword, and WITHOUT a sign bit to fill a bit may stay where it is
// inside its nibble: FP4 E2M1 reads nibble bit 0 as 0.5, bit 1 as 1.0, bit 2 as 2.0, so a candidate fragment is three ANDs
// (of the complement) and one shift + AND for the bits at position 3 (-> position 2) -- 5 instructions per 32 bits between
// two groups of products where the +-1 form has 7.  The query side (expanded once per kernel) carries the weights that make
// every product 2: 4.0 (0x6), 2.0 (0x4), 1.0 (0x2), 1.0.
__device__ __forceinline__ v8i bf_expand_nn_cand(uint32_t w) {
    const uint32_t n = ~w;
    return (v8i){(int)(n & 0x11111111u), (int)(n & 0x22222222u), (int)(n & 0x44444444u), (int)(~(w >> 1) & 0x44444444u), 0, 0, 0, 0};
}
__device__ __forceinline__ v8i bf_expand_nn_query(uint32_t w) {
    const uint32_t b0 = w & 0x11111111u;
    return (v8i){(int)((b0 << 1) | (b0 << 2)), (int)((w & 0x22222222u) << 1), (int)((w & 0x44444444u) >> 1), (int)((w >> 2) & 0x22222222u), 0, 0, 0, 0};
}
__device__ __forceinline__ int popc128(v4u x) { return __popc(x.x) + __popc(x.y) + __popc(x.z) + __popc(x.w); }

// PIPE: 2 the sign-free form (above; the default for large batches), 0 plain, 1 with s_setprio around the products (+3..4 % sustained: the matrix pipe idles less), 10-12 timing diagnostics
template <bool WANT_DIST, int NT, int PIPE = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void bf2nn_mfma4r_kernel(const FrameDev *__restrict__ table,
                                                           const int32_t *__restrict__ fa, const int32_t *__restrict__ fb,
                                                           int blocks_per_pair, float ratio, int th, int track_below, int stride,
                                                           int32_t *__restrict__ match, int32_t *__restrict__ best_d,
                                                           int32_t *__restrict__ second_d, int32_t *__restrict__ counts) {
    constexpr bool NN = PIPE == 2 || PIPE == 4;
    // DEF (PIPE 4): the last 64 bits of a comparison are DEFERRED.  A block's results after three of the four products are
    // partial distances over 192 bits (+ the query's popcount over the same bits): lower bounds of the full ones, since the
    // fourth product only adds.  They are ranked against the bound as usual, and only when some lane finds a partial result
    // below its bound does the wave issue the fourth group of products, add the rows' last-64-bit popcounts and rank again.
    // For pairs that share nothing (distance 128 +- 8, 192-bit part 96 +- 7, bound 67) that happens in about one block of
    // twenty: a quarter of the products and of the expansion is not executed, results are identical.
    constexpr bool DEF = PIPE == 4;
    extern __shared__ float s_pc[];    // NN: popcount of every candidate row (DEF: of its first 192 bits; 1024 for the padding rows), the accumulators' initial values
    const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
    const int p = bid / blocks_per_pair;
    const int qb = bid - p * blocks_per_pair;
    const FrameG A(table[fa[p]]), B(table[fb[p]]);
    const int nA = A.n, nB = B.n;
    constexpr int QW = 32 * NT;
    if (qb * (4 * QW) >= nA) {
        for (int i = threadIdx.x; i < 4 * QW; i += 256) if (qb * (4 * QW) + i < stride) match[(size_t)p * stride + qb * (4 * QW) + i] = -1;
        return;
    }
    const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63, col = l & 31, h = l >> 5;
    const int qbase = qb * (4 * QW) + wave * QW;
    // LDS: s_pc[nB8] | s_pq[4][256] (query popcounts: full | last 64 bits << 16) | DEF: s_tf[4][256] (the full bound as float
    // bits) | DEF: s_bq3[4][256] x 16 B | DEF: s_pl[nB8] bytes (popcount of a candidate row's last 64 bits = dword 3 of both halves)
    int *const s_pq = (int *)(s_pc + ((nB + 31) & ~31)) + tid;
    int *const s_tf = s_pq + 4 * 256;
    v4u *const s_bq3 = (v4u *)(s_pq - tid + 8 * 256) + tid;            // DEF: the lane's query fragments of the fourth product, [4][256] x 16 B (only the rare path reads them: 16 registers less in the loop)
    uint8_t *const s_pl = (uint8_t *)(s_pq - tid + 8 * 256 + 4 * 256 * 4);
    if (NN) {                          // the one barrier of this form: every wave of the workgroup, with or without queries
        const gptr<const v4u> rows = (gptr<const v4u>)B.desc;
        for (int r = tid; r < ((nB + 31) & ~31); r += 256) {
            float v = 1024.f;
            int last = 0;
            if (r < nB) {
                const v4u x0 = rows[2 * (size_t)r], x1 = rows[2 * (size_t)r + 1];
                last = DEF ? __popc(x0.w) + __popc(x1.w) : 0;
                v = (float)(popc128(x0) + popc128(x1) - last);
            }
            s_pc[r] = v;
            if (DEF) s_pl[r] = (uint8_t)last;
        }
        __syncthreads();
    }
    if (qbase >= nA) {                 // a wave without queries simply leaves (no barrier below this line)
        for (int i = l; i < QW; i += 64) if (qbase + i < stride) match[(size_t)p * stride + qbase + i] = -1;
        return;
    }
    const long long diag_c0 = PIPE == 10 ? clock64() : 0, diag_w0 = PIPE == 10 ? wall_clock64() : 0;
    uint32_t ones = 0x22222222u;
    asm volatile("" : "+v"(ones));
    v8i bq[NT][DEF ? 3 : 4];
    // NN: the popcount of the lane's query of tile t, kept in LDS (read where a bound changes and at the end only; four more
    // registers would cost the third wave per SIMD).  best / second hold (d + pq) << 16 | candidate in this form.
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int row = min(qbase + t * 32 + col, max(nA - 1, 0));
        const v4u x = ((gptr<const v4u>)A.desc)[2 * (size_t)row + h];
        const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int s = 0; s < (DEF ? 3 : 4); s++) bq[t][s] = NN ? bf_expand_nn_query(w[s]) : bf_expand_fp4(w[s], ones);
        if (DEF) { const v8i f = bf_expand_nn_query(w[3]); s_bq3[256 * t] = (v4u){(uint32_t)f[0], (uint32_t)f[1], (uint32_t)f[2], (uint32_t)f[3]}; }
        if (NN) {
            const int half = popc128(x) | (DEF ? __popc(x.w) << 16 : 0);         // both counts at once: neither reaches 2^16
            s_pq[256 * t] = half + __shfl_xor(half, 32);
        }
    }
    uint32_t best[NT], second[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        best[t] = ((256u + (NN ? (uint32_t)s_pq[256 * t] & 0xFFFFu : 0u)) << 16) | 0xFFFFu;
        second[t] = best[t];
    }
    const int one = 0x7F7F7F7F;
    const gptr<const v4u> rowsB = (gptr<const v4u>)B.desc;
    const int nB8 = (nB + 31) & ~31;
    auto fetch = [&](int rb) -> v4u { return rowsB[2 * (size_t)min(rb + col, max(nB - 1, 0)) + h]; };   // rows past the end: the last row again (never ranked)
    // bound per tile as integer bits; a bound below zero (more than 128 differing bits still relevant: only until the second
    // best falls below that) sends every block through the exact path
    int Ti[NT];
    bool anyneg = false;
    auto set_bounds = [&]() {
        bool neg = false;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            if (NN) {                  // a result is d + pq >= 0: relevant below bound + pq, nothing is ever negative
                const uint32_t pk = (uint32_t)s_pq[256 * t];
                const uint32_t tfull = min(second[t] >> 16, (uint32_t)track_below + (pk & 0xFFFFu));
                Ti[t] = __float_as_int((float)(tfull - (pk >> 16)));           // DEF: the bound of the 192-bit part (pk >> 16 is 0 otherwise)
                if (DEF) s_tf[256 * t] = __float_as_int((float)tfull);
                continue;
            }
            const float Tf = (float)(256 - 2 * (int)min(second[t] >> 16, (uint32_t)track_below));
            Ti[t] = Tf >= 0.f ? __float_as_int(Tf) : (int)0x80000000;
            neg = neg || Tf < 0.f;
        }
        anyneg = !NN && __any(neg);
    };
    set_bounds();
    // NL = the wave's tiles that hold queries (wave-uniform; the last wave of a 2000-row frame has three of four): tiles
    // without queries cost no products
    auto scan = [&](auto nl_tag) {
    constexpr int NL = decltype(nl_tag)::value;
    v4u nxt = nB8 > 0 ? fetch(0) : v4u{0, 0, 0, 0};
    v4u nxt2 = nB8 > 32 ? fetch(32) : nxt;          // candidate rows are requested two blocks ahead
    // NN: register r of a result is candidate rb + 4 h + (r & 3) + 8 (r >> 2): four 16-byte LDS reads, the same address in
    // every lane of a half
    auto load_cinit = [&](int rb) -> v16f {
        const v4f *cp = (const v4f *)(s_pc + rb + 4 * h);
        v16f c;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const v4f x = cp[2 * j];
            c[4 * j] = x.x; c[4 * j + 1] = x.y; c[4 * j + 2] = x.z; c[4 * j + 3] = x.w;
        }
        return c;
    };
    v16f cnext = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (NN && nB8 > 0) cnext = load_cinit(0);
    for (int rb = 0; rb < nB8; rb += 32) {
        if (PIPE == 1 || NN || PIPE == 3) __builtin_amdgcn_s_setprio(1);       // a wave that has matrix instructions to issue goes before waves that are ranking
        const v4u cur = nxt;
        nxt = nxt2;
        if (rb + 64 < nB8) nxt2 = fetch(rb + 64);
        const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
        v16f acc[NT];
        v16f cinit = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (NN) cinit = cnext;         // in the last tile's registers: its first product overwrites them in place
#pragma unroll
        for (int s = 0; s < (DEF ? 3 : 4); s++) {
            const v8i a = PIPE == 11 || PIPE == 12 ? (v8i){(int)w[0], (int)w[1], (int)w[2], (int)w[3], 0, 0, 0, 0}                // 10-12: timing diagnostics (results are wrong)
                        : NN ? bf_expand_nn_cand(w[s]) : bf_expand_fp4(w[s], ones);
#pragma unroll
            for (int t = 0; t < NL; t++)
                acc[t] = PIPE == 3 ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bq[t][s], s == 0 ? cinit : acc[t], 4, 4, 0, one, 0, one)   // A/B: the scaled encoding
                                   : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bq[t][s], s == 0 ? cinit : acc[t], 4, 4, 0, 0, 0, 0);
        }
        if (PIPE == 10 || PIPE == 12) {
#pragma unroll
            for (int t = 0; t < NL; t++) {
#ifdef __HIP_DEVICE_COMPILE__
                asm volatile("" ::"v"(acc[t]));
#endif
            }
            continue;
        }
        if (PIPE == 1 || NN || PIPE == 3) __builtin_amdgcn_s_setprio(0);
        // dot products are ranked as integers (a positive float orders like its bits, every negative float is a negative
        // integer): eight v_max3_i32 (v_min3_i32 in the sign-free form) per tile fold its sixteen values and the bound
        auto fold = [&](auto t0c, auto t1c, auto full) -> bool {
            bool fire = false;
#pragma unroll
            for (int t = decltype(t0c)::value; t < decltype(t1c)::value; t++) {
                int m;
                const int Tb = decltype(full)::value ? s_tf[256 * t] : Ti[t];       // the full bound is read where it is used (rare path: no registers held for it)
                if (NN) {                  // smallest of sixteen non-negative values and the bound
                    m = min(min(__float_as_int(acc[t][0]), __float_as_int(acc[t][1])), Tb);
#pragma unroll
                    for (int r = 2; r < 16; r += 2) m = min(min(m, __float_as_int(acc[t][r])), __float_as_int(acc[t][r + 1]));
                } else {
                    m = max(max(__float_as_int(acc[t][0]), __float_as_int(acc[t][1])), Tb);
#pragma unroll
                    for (int r = 2; r < 16; r += 2) m = max(max(m, __float_as_int(acc[t][r])), __float_as_int(acc[t][r + 1]));
                }
                fire = fire || m != Tb;
            }
            return fire;
        };
        constexpr std::false_type part{};
        constexpr std::true_type whole{};
        auto exact = [&](auto t0c, auto t1c) {
            const int cand0 = rb + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int cand = cand0 + (r & 3) + 8 * (r >> 2);
#pragma unroll
                for (int t = decltype(t0c)::value; t < decltype(t1c)::value; t++) {
                    const int d = NN ? (int)acc[t][r] : (256 - (int)acc[t][r]) >> 1;
                    const uint32_t key = cand < nB ? ((uint32_t)d << 16) | (uint32_t)cand : 0xFFFFFFFFu;
                    second[t] = umed3(best[t], second[t], key); best[t] = min(best[t], key);
                }
            }
            set_bounds();
        };
        constexpr std::integral_constant<int, 0> c0{};
        constexpr std::integral_constant<int, NL> cn{};
        if (DEF) {
            if (__any(fold(c0, cn, part))) {
                // some 192-bit part is below its bound: the last 64 bits after all, for the whole block.  First the rows'
                // last-64-bit popcounts onto the results at hand (integers: the order of the additions does not matter), then
                // the fourth group of products on top
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t pl4 = *(const uint32_t *)(s_pl + rb + 4 * h + 8 * j);
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const float add = (float)((pl4 >> (8 * i)) & 255u);
#pragma unroll
                        for (int t = 0; t < NL; t++) acc[t][4 * j + i] += add;
                    }
                }
                const v8i a = bf_expand_nn_cand(w[3]);
#pragma unroll
                for (int t = 0; t < NL; t++) {
                    const v4u f = s_bq3[256 * t];
                    const v8i b3 = {(int)f.x, (int)f.y, (int)f.z, (int)f.w, 0, 0, 0, 0};
                    acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b3, acc[t], 4, 4, 0, 0, 0, 0);
                }
                if (__any(fold(c0, cn, whole))) exact(c0, cn);
            }
        } else if (anyneg || __any(fold(c0, cn, part))) exact(c0, cn);
        // the next block's initial values, requested as soon as this block's results are dead: the reads complete under the
        // loop's address arithmetic and the first expansion instead of in front of the first product
        if (NN) cnext = load_cinit(min(rb + 32, nB8 - 32));
    }
    };
    const int nl = min(NT, (nA - qbase + 31) >> 5);
    if (nl == NT) scan(std::integral_constant<int, NT>{});
    else if (NT > 2 && nl == 3) scan(std::integral_constant<int, (NT > 2 ? 3 : 1)>{});
    else if (NT > 1 && nl == 2) scan(std::integral_constant<int, (NT > 1 ? 2 : 1)>{});
    else scan(std::integral_constant<int, 1>{});
    if (PIPE == 10) {          // diagnostic: the shader clock this wave saw, in MHz (wall_clock64 ticks at 100 MHz)
        const long long dc = clock64() - diag_c0, dw = wall_clock64() - diag_w0;
        if (l == 0 && wave == 0 && qb == 0) match[(size_t)p * stride] = (int32_t)(dc * 100 / max(dw, 1LL));
        return;
    }
    int accepted = 0;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const uint32_t pb = (uint32_t)__shfl_xor((int)best[t], 32), ps = (uint32_t)__shfl_xor((int)second[t], 32);
        const uint32_t b = min(best[t], pb), s2 = min(max(best[t], pb), min(second[t], ps));
        const int qi = qbase + t * 32 + col;
        if (h == 0 && qi < nA) {
            const int off = NN ? s_pq[256 * t] & 0xFFFF : 0;
            const int bd = (int)(b >> 16) - off, sd = (int)(s2 >> 16) - off;
            const int bidx = bd >= 256 ? -1 : (int)(b & 0xFFFFu);
            int mm = -1;
            if (bd <= th && (float)bd < __fmul_rn(ratio, (float)sd)) { mm = bidx; accepted++; }  // :304-308
            const size_t o = (size_t)p * stride + qi;
            match[o] = mm;
            if (WANT_DIST) {
                if (best_d) best_d[o] = bd;
                if (second_d) second_d[o] = sd;
            }
        } else if (h == 0 && qi < stride) {
            match[(size_t)p * stride + qi] = -1;
        }
    }
    for (int off = 32; off > 0; off >>= 1) accepted += __shfl_down(accepted, off, 64);
    if (l == 0 && accepted) atomicAdd(&counts[p], accepted);
}

int g_bf_variant = 0;

}  // namespace

// tuning knob used by tests/ and scripts/ (not part of the public ABI).  Every form selectable in the release library returns
// the same matches; the timing diagnostics that do not (4, 18-20, ORBM_MFMA_DIAG) exist only in builds with -DORBM_DIAG
// (`make diag` -> liborbmatch_hip_diag.so, for scripts/), and a request for one of them selects the default form here.
extern "C" void orbm__bf2nn_variant(int v) {
#ifndef ORBM_DIAG
    if (v == 4 || (v >= 18 && v <= 20)) v = 0;
#endif
    g_bf_variant = v;
}

// 0 = default (15): the FP4 MFMA form fed from the raw descriptor rows, query tiles per wave chosen by batch size (12: two,
// 13: four, 14: one); 10 = the same product fed from pre-expanded FP4 images through LDS (ORBMATCH_BF2NN_KERNEL=lds);
// 9 = int8 MFMA; 11 = the vector-ALU kernel (ORBMATCH_BF2NN_KERNEL=valu); the rest are A/B and diagnostic forms
static int bf2nn_variant_in_use() {
    // opt-in at run time: ORBMATCH_BF2NN_EARLY_ABANDON=1 selects the early-abandon kernel (same results)
    static const bool env_early_abandon = [] { const char *e = getenv("ORBMATCH_BF2NN_EARLY_ABANDON"); return e && e[0] == '1'; }();
    // ORBMATCH_BF2NN_KERNEL=valu selects the vector-ALU xor + popcount kernel (no +-1 image of the frames is built)
    static const bool env_valu = [] { const char *e = getenv("ORBMATCH_BF2NN_KERNEL"); return e && !strcmp(e, "valu"); }();
    static const int env_form = [] { const char *e = getenv("ORBMATCH_BF2NN_KERNEL"); return !e ? 0 : !strcmp(e, "lds") ? 10 : !strcmp(e, "raw2") ? 12 : !strcmp(e, "raw4") ? 13 : !strcmp(e, "full") ? 22 : 0; }();      // full: every bit of every comparison multiplied (no deferred last 64 bits)
    return g_bf_variant != 0 ? g_bf_variant : env_early_abandon ? 8 : env_valu ? 11 : env_form ? env_form : 15;
}

extern "C" int orbm_bf2nn_prepare(orbm_ctx *ctx) {
    if (!ctx) return ORBM_EINVAL;
    const int variant = bf2nn_variant_in_use();
    if (variant != 9 && variant != 10) return ORBM_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return orbm_ensure_image(ctx, variant == 9 ? 8 : 4, 0, nullptr);
}

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
    const int variant = bf2nn_variant_in_use();
    if (variant == 9 || variant == 10) {    // MFMA forms: +-1 images of the frames, built on first use (before any staging: it uses the workspace)
        const int bits = variant == 9 ? 8 : 4;
        int rc8 = orbm_ensure_image(ctx, bits, io_on_device ? 0 : n_pairs, io_on_device ? nullptr : frame_a);
        if (!rc8 && !io_on_device) rc8 = orbm_ensure_image(ctx, bits, n_pairs, frame_b);
        if (rc8) return rc8;
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
    // smallest distance that can never influence the accept decision (see kernel header); 257 = track all
    int track_below = 257;
    if (!best_d && !second_d && g_bf_variant != 7) {
        for (int t = std::max(th + 1, 1); t <= 256; t++)
            if ((float)th < ratio * (float)t) { track_below = t; break; }
    }
    if (variant == 10) {                // MX-scaled FP4 MFMA form
        const int per_block = 4 * 32 * MF4_NT;
        const int bpp9 = (max_na + per_block - 1) / per_block;
        const long long grid9 = (long long)n_pairs * bpp9;
        if (grid9 > 0x7FFFFFFFLL) return orbm_fail(ctx, ORBM_ELIMIT, "bf2nn_batch: grid too large");
#define MF4_LAUNCH(WD, DG)                                                                                                       \
        hipLaunchKernelGGL((bf2nn_mfma4_kernel<WD, DG, MF4_NT>), dim3((unsigned)grid9), dim3(256), 0, ctx->stream, ctx->d_frames, (const int32_t *)dfa, \
                           (const int32_t *)dfb, bpp9, ratio, th, track_below, stride, (int32_t *)dm, (int32_t *)dbd, (int32_t *)dsd, (int32_t *)dc)
#ifdef ORBM_DIAG
        static const int diag10 = [] { const char *e = getenv("ORBM_MFMA_DIAG"); return e ? atoi(e) : 0; }();
        if (diag10 == 1) MF4_LAUNCH(false, 1);
        else if (diag10 == 2) MF4_LAUNCH(false, 2);
        else
#endif
        if (best_d || second_d) MF4_LAUNCH(true, 0);
        else MF4_LAUNCH(false, 0);
#undef MF4_LAUNCH
        HIPCHK(ctx, hipGetLastError());
        return io.finish();
    }
    if (variant >= 12 && variant <= 24) {     // FP4 MFMA from the raw rows, no LDS (12: two query tiles per wave, 13: four, 14: one)
        // 15: by batch size (scripts/bf_small_perf.py).  Four tiles per wave amortise the expansion of a candidate fragment over
        // 16 MFMAs (3 waves per SIMD) and win from about a thousand 2000 x 2000 pairs; below that more, smaller
        // workgroups fill the chip sooner: two tiles from 32 pairs, one tile for a handful (one pair alone: 24 us).
        int nt = variant == 12 ? 2 : variant == 13 || variant >= 16 ? 4 : 1;      // 16-17: A/B forms of four tiles
        if (variant == 15) {
            const long long wg4 = (long long)n_pairs * ((max_na + 511) / 512), wg2 = (long long)n_pairs * ((max_na + 255) / 256);
            nt = wg4 >= 3000 ? 4 : wg2 >= 256 ? 2 : 1;
        }
        const int per_block = 4 * 32 * nt;
        const int bpp = (max_na + per_block - 1) / per_block;
        const long long gridr = (long long)n_pairs * bpp;
        if (gridr > 0x7FFFFFFFLL) return orbm_fail(ctx, ORBM_ELIMIT, "bf2nn_batch: grid too large");
        // sign-free operands (PIPE 2) for the large batches: the candidate frame's row popcounts sit in LDS, 4 bytes per row.
        // The pair lists may live on the device, so the bound is the largest live frame of the context.
        size_t nn_lds = 0;
        if (nt == 4 && (variant == 15 || variant == 22 || variant == 24)) {      // 24: deferred form whatever the batch size (if the bound allows)
            int max_nb = 0;
            if (!io_on_device) for (int p = 0; p < n_pairs; p++) max_nb = std::max(max_nb, ctx->h_frames[frame_b[p]].n);
            else for (size_t i = 0; i < ctx->h_frames.size(); i++) if (ctx->frame_arena[i] >= 0) max_nb = std::max(max_nb, ctx->h_frames[i].n);
            if (max_nb <= 8192) nn_lds = (size_t)((max_nb + 31) & ~31) * 5 + 2 * 256 * 4 * 4 + 256 * 4 * 16;      // + the queries' popcounts and full bounds, their fourth fragments, the rows' last-64-bit counts
        }
#define MF4R_LAUNCH(WD, NTT, ...)                                                                                                \
        hipLaunchKernelGGL((bf2nn_mfma4r_kernel<WD, NTT, ##__VA_ARGS__>), dim3((unsigned)gridr), dim3(256), lds, ctx->stream, ctx->d_frames, (const int32_t *)dfa, \
                           (const int32_t *)dfb, bpp, ratio, th, track_below, stride, (int32_t *)dm, (int32_t *)dbd, (int32_t *)dsd, (int32_t *)dc)
        hipEvent_t *tev = orbm_timing_slot(ctx);           // orbm_kernel_timing: this launch alone, on its stream
        if (tev) HIPCHK(ctx, hipEventRecord(tev[0], ctx->stream));
        size_t lds = 0;
        if (variant == 13) { if (best_d || second_d) MF4R_LAUNCH(true, 4); else MF4R_LAUNCH(false, 4); }     // A/B: without s_setprio
        else if (variant == 16) { if (best_d || second_d) MF4R_LAUNCH(true, 4, 3); else MF4R_LAUNCH(false, 4, 3); }
#ifdef ORBM_DIAG
        else if (variant == 18) MF4R_LAUNCH(false, 4, 10);
        else if (variant == 19) MF4R_LAUNCH(false, 4, 11);
        else if (variant == 20) MF4R_LAUNCH(false, 4, 12);
#endif
        else if (nt == 2) { if (best_d || second_d) MF4R_LAUNCH(true, 2); else MF4R_LAUNCH(false, 2); }
        else if (nt == 1) { if (best_d || second_d) MF4R_LAUNCH(true, 1); else MF4R_LAUNCH(false, 1); }
        // the deferred form pays while a lower bound over 192 bits (96 +- 7 for unrelated descriptors) rarely reaches the
        // tracking bound: matches only, and an acceptance rule whose bound stays below 72 (th 50 at ratio 0.75: 67)
        else if (nn_lds && !best_d && !second_d && track_below <= 72 && variant != 22) { lds = nn_lds; MF4R_LAUNCH(false, 4, 4); }
        else if (nn_lds) { lds = nn_lds; if (best_d || second_d) MF4R_LAUNCH(true, 4, 2); else MF4R_LAUNCH(false, 4, 2); }
        else { if (best_d || second_d) MF4R_LAUNCH(true, 4, 1); else MF4R_LAUNCH(false, 4, 1); }     // 17 (and frames above 8192 rows): the +-1 form with four tiles; 22: the sign-free form whatever the batch size
#undef MF4R_LAUNCH
        HIPCHK(ctx, hipGetLastError());
        if (tev) HIPCHK(ctx, hipEventRecord(tev[1], ctx->stream));
        return io.finish();
    }
    if (variant == 9) {                 // MFMA form: +-1 int8 images of the frames, built on first use
        const int bpp9 = (max_na + 255) / 256;
        const long long grid9 = (long long)n_pairs * bpp9;
        if (grid9 > 0x7FFFFFFFLL) return orbm_fail(ctx, ORBM_ELIMIT, "bf2nn_batch: grid too large");
#ifdef ORBM_DIAG
        static const int diag9 = [] { const char *e = getenv("ORBM_MFMA_DIAG"); return e ? atoi(e) : 0; }();
        if (diag9 == 1)
            hipLaunchKernelGGL((bf2nn_mfma_kernel<false, 1>), dim3((unsigned)grid9), dim3(256), 0, ctx->stream, ctx->d_frames, (const int32_t *)dfa,
                               (const int32_t *)dfb, bpp9, ratio, th, track_below, stride, (int32_t *)dm, (int32_t *)dbd, (int32_t *)dsd, (int32_t *)dc);
        else if (diag9 == 2)
            hipLaunchKernelGGL((bf2nn_mfma_kernel<false, 2>), dim3((unsigned)grid9), dim3(256), 0, ctx->stream, ctx->d_frames, (const int32_t *)dfa,
                               (const int32_t *)dfb, bpp9, ratio, th, track_below, stride, (int32_t *)dm, (int32_t *)dbd, (int32_t *)dsd, (int32_t *)dc);
        else
#endif
        if (best_d || second_d)
            hipLaunchKernelGGL((bf2nn_mfma_kernel<true, 0>), dim3((unsigned)grid9), dim3(256), 0, ctx->stream, ctx->d_frames, (const int32_t *)dfa,
                               (const int32_t *)dfb, bpp9, ratio, th, track_below, stride, (int32_t *)dm, (int32_t *)dbd, (int32_t *)dsd, (int32_t *)dc);
        else
            hipLaunchKernelGGL((bf2nn_mfma_kernel<false, 0>), dim3((unsigned)grid9), dim3(256), 0, ctx->stream, ctx->d_frames, (const int32_t *)dfa,
                               (const int32_t *)dfb, bpp9, ratio, th, track_below, stride, (int32_t *)dm, (int32_t *)dbd, (int32_t *)dsd, (int32_t *)dc);
        HIPCHK(ctx, hipGetLastError());
        return io.finish();
    }
    const int R = (variant == 3 || variant >= 7 || variant <= 0) ? 2 : 4;
    const int bpp = (max_na + 256 * R - 1) / (256 * R);
    const long long grid = (long long)n_pairs * bpp;
    if (grid > 0x7FFFFFFFLL) return orbm_fail(ctx, ORBM_ELIMIT, "bf2nn_batch: grid too large");
#define BF_LAUNCH(RR, UU, DD)                                                                                   \
    hipLaunchKernelGGL((bf2nn_kernel<RR, UU, DD>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, ctx->d_frames, \
                       (const int32_t *)dfa, (const int32_t *)dfb, bpp, ratio, th, track_below, stride, (int32_t *)dm, \
                       (int32_t *)dbd, (int32_t *)dsd, (int32_t *)dc)
    switch (variant) {
        case 1: BF_LAUNCH(4, 2, 0); break;
        case 2: BF_LAUNCH(4, 1, 0); break;
        case 3: BF_LAUNCH(2, 2, 0); break;
#ifdef ORBM_DIAG
        case 4: BF_LAUNCH(4, 2, 1); break;   // diagnostic: no exact updates (results wrong)
#endif
        case 5: BF_LAUNCH(4, 2, 2); break;   // diagnostic: exact update on every group
        case 6: BF_LAUNCH(4, 4, 0); break;
        case 8: BF_LAUNCH(BF_R, BF_U, 3); break;   // early abandon (opt-in)
        default: BF_LAUNCH(BF_R, BF_U, 0); break;
    }
#undef BF_LAUNCH
    HIPCHK(ctx, hipGetLastError());
    return io.finish();
}
