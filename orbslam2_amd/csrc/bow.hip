// bow.hip -- BoW-node-grouped matching: SearchByBoW (a7: ORBmatcher.cc:218-385, a8: :703-854) and
// SearchForTriangulation (a9: :868-1093), batched over keyframe pairs.
//
// One 256-thread workgroup per pair.  A feature index lives in exactly one node of its FeatureVector,
// so the reference's "already matched" claims (vpMapPointMatches[idxF] / vbMatched2) only couple queries
// of the SAME node: (pair, node) groups are independent.  Each wavefront takes common nodes (found by
// binary search of the a-side node id in the b-side node array == the lower_bound merge walk of
// :248-352), walks that node's queries IN ORDER, spreads the node's candidates over its 64 lanes and
// reduces (best, second) over lanes with key = (dist << 16) | position-in-node so that the lowest
// position wins ties exactly like the reference's strict-< scan.  The first 64 candidates of a node
// stay in registers for all of its queries.  Ratio test, rotation histogram, ComputeThreeMaxima and
// the erase pass run in the same launch.
#include <string.h>
#include <algorithm>
#include <mutex>
#include "orbm_internal.h"

namespace {

constexpr uint32_t INITKEY = (256u << 16) | 0xFFFFu;
constexpr int MAX_N = 65536;
constexpr int BOWQ_STAGE = 128;   // candidate rows a wave keeps in LDS for one group of queries

__device__ __forceinline__ void load_row(gptr<const uint32_t> desc, int idx, uint32_t (&r)[8]) {
    const gptr<const v4u> p = (gptr<const v4u>)desc + 2 * (size_t)idx;
    const v4u lo = p[0], hi = p[1];
    r[0] = lo.x; r[1] = lo.y; r[2] = lo.z; r[3] = lo.w; r[4] = hi.x; r[5] = hi.y; r[6] = hi.z; r[7] = hi.w;
}

// merge two sorted (best <= second) key pairs across lanes; after the butterfly every lane holds the result
__device__ __forceinline__ void wave_top2(uint32_t &best, uint32_t &second) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t ob = __shfl_xor(best, off, 64), os = __shfl_xor(second, off, 64);
        const uint32_t nb = min(best, ob);
        second = min(max(best, ob), min(second, os));
        best = nb;
    }
}

__device__ __forceinline__ uint32_t wave_min(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
    return v;
}

__device__ __forceinline__ int find_node(gptr<const uint32_t> nodes, int n, uint32_t key) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (nodes[mid] < key) lo = mid + 1; else hi = mid;
    }
    return (lo < n && nodes[lo] == key) ? lo : -1;
}

// MODE 0: a7 (KF -> F), 1: a8 (KF -> KF), 2: a9 (triangulation)
// SPLIT = false: one workgroup per pair does everything (large batches: the chip is full anyway).
// SPLIT = true : `splits` workgroups share a pair's nodes (a single SearchByBoW call would otherwise run on one CU);
//                match rows are pre-set to -1 by the host side, histogram and count go to `pair_acc` (32 ints per pair)
//                and bow_finish_kernel applies the rotation filter once all of them are done.
template <int MODE, bool SPLIT>
__global__ __launch_bounds__(256) void bow_kernel(const FrameDev *__restrict__ table, const int32_t *__restrict__ fa,
                                                  const int32_t *__restrict__ fb, float ratio, int check_ori,
                                                  int only_stereo, const float *__restrict__ f12all,
                                                  const float *__restrict__ epiall, int stride,
                                                  int32_t *__restrict__ match_all, int32_t *__restrict__ counts,
                                                  int splits, int *__restrict__ pair_acc) {
    __shared__ uint32_t claimed[MAX_N / 32];  // bit per b-side keypoint (a7: vpMapPointMatches set, a8: vbMatched2)
    __shared__ int hist[ORBM_HISTO_LENGTH];
    __shared__ int keep[3];
    __shared__ int nmatches;
    const int p = SPLIT ? (int)blockIdx.x / splits : (int)xcd_remap(blockIdx.x, gridDim.x);
    const int part = SPLIT ? (int)blockIdx.x - p * splits : 0;
    const int nparts = SPLIT ? splits : 1;
    const FrameG A(table[fa[p]]), B(table[fb[p]]);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nOut = MODE == 0 ? B.n : A.n;
    int32_t *match = match_all + (size_t)p * stride;
    if (!SPLIT) for (int i = tid; i < nOut; i += 256) match[i] = -1;
    if (MODE != 2)
        for (int i = tid; i < (B.n + 31) / 32; i += 256) claimed[i] = 0;
    if (tid < ORBM_HISTO_LENGTH) hist[tid] = 0;
    if (tid == 0) nmatches = 0;
    __syncthreads();

    const gptr<const float> angB = MODE == 0 ? B.angle_raw : B.angle;  // :323 reads F.mvKeys on the frame side
    float F12[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    float ex = 0.f, ey = 0.f;
    if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 9; i++) F12[i] = f12all[(size_t)p * 9 + i];
        ex = epiall[2 * p]; ey = epiall[2 * p + 1];
    }

    int my_matches = 0;  // wave-uniform, counted on lane 0
    for (int ia = part * 4 + wave; ia < A.n_nodes; ia += 4 * nparts) {
        const uint32_t node_id = A.fv_node[ia];
        const int guess = min(ia, B.n_nodes - 1);                           // usually the same index on both sides (see bowq_kernel)
        const int ib = (guess >= 0 && B.fv_node[guess] == node_id) ? guess : find_node(B.fv_node, B.n_nodes, node_id);
        if (ib < 0) continue;
        const uint32_t offA = A.fv_off[ia], nA = A.fv_off[ia + 1] - offA;
        const uint32_t offB = B.fv_off[ib], nB = B.fv_off[ib + 1] - offB;
        if (nA == 0 || nB == 0) continue;
        // first chunk of candidates stays in registers for every query of the node
        uint32_t c0[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int idxB0 = -1;
        bool ok0 = false;           // static candidate gates
        float x2 = 0, y2 = 0; int oct2 = 0; bool st2 = false;
        if ((uint32_t)lane < nB) {
            idxB0 = (int)B.fv_idx[offB + lane];
            load_row(B.desc, idxB0, c0);
            const uint8_t fl = B.mp_flags[idxB0];
            if (MODE == 0) ok0 = true;
            if (MODE == 1) ok0 = (fl & ORBM_MP_PRESENT) && !(fl & ORBM_MP_BAD);          // :766-770
            if (MODE == 2) {
                st2 = B.uright[idxB0] >= 0;
                ok0 = !(fl & ORBM_MP_PRESENT) && (!only_stereo || st2);                  // :970, :979-981
                x2 = B.x[idxB0]; y2 = B.y[idxB0]; oct2 = B.octave[idxB0];
            }
        }
        uint32_t lq[8];
        int lq_idx = 0, lq_st = 0;
        uint32_t lq_flags = 0;
        float lq_x = 0.f, lq_y = 0.f;
        for (uint32_t qa = 0; qa < nA; qa++) {
            // The node's queries are fetched 64 at a time, one per lane (index, flags, descriptor, position), and
            // handed to the whole wave with v_readlane: the in-order loop then has no dependent global load.
            if ((qa & 63u) == 0) {
                const uint32_t ql = qa + lane;
                lq_idx = 0; lq_flags = 0; lq_x = 0.f; lq_y = 0.f; lq_st = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) lq[i] = 0;
                if (ql < nA) {
                    lq_idx = (int)A.fv_idx[offA + ql];
                    lq_flags = A.mp_flags[lq_idx];
                    load_row(A.desc, lq_idx, lq);
                    if (MODE == 2) { lq_x = A.x[lq_idx]; lq_y = A.y[lq_idx]; lq_st = A.uright[lq_idx] >= 0 ? 1 : 0; }
                }
            }
            const int src = (int)(qa & 63u);
            const int idxA = __builtin_amdgcn_readlane(lq_idx, src);
            const uint32_t fA = (uint32_t)__builtin_amdgcn_readlane((int)lq_flags, src);
            bool st1 = false;
            if (MODE == 2) {
                if (fA & ORBM_MP_PRESENT) continue;                                       // :932
                st1 = __builtin_amdgcn_readlane(lq_st, src) != 0;
                if (only_stereo && !st1) continue;                                        // :939-941
            } else {
                if (!(fA & ORBM_MP_PRESENT) || (fA & ORBM_MP_BAD)) continue;              // :263-267 / :746-749
            }
            uint32_t q[8];
#pragma unroll
            for (int i = 0; i < 8; i++) q[i] = (uint32_t)__builtin_amdgcn_readlane((int)lq[i], src);
            const float x1 = MODE == 2 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lq_x), src)) : 0.f;
            const float y1 = MODE == 2 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lq_y), src)) : 0.f;
            uint32_t best = INITKEY, second = INITKEY;
            for (uint32_t cb = 0; cb < nB; cb += 64) {
                const uint32_t c = cb + lane;
                if (c >= nB) continue;
                uint32_t cd[8];
                int idxB; bool ok; float cx2, cy2; int coct2; bool cst2;
                if (cb == 0) {
#pragma unroll
                    for (int i = 0; i < 8; i++) cd[i] = c0[i];
                    idxB = idxB0; ok = ok0; cx2 = x2; cy2 = y2; coct2 = oct2; cst2 = st2;
                } else {
                    idxB = (int)B.fv_idx[offB + c];
                    load_row(B.desc, idxB, cd);
                    const uint8_t fl = B.mp_flags[idxB];
                    ok = true; cx2 = cy2 = 0; coct2 = 0; cst2 = false;
                    if (MODE == 1) ok = (fl & ORBM_MP_PRESENT) && !(fl & ORBM_MP_BAD);
                    if (MODE == 2) {
                        cst2 = B.uright[idxB] >= 0;
                        ok = !(fl & ORBM_MP_PRESENT) && (!only_stereo || cst2);
                        cx2 = B.x[idxB]; cy2 = B.y[idxB]; coct2 = B.octave[idxB];
                    }
                }
                if (!ok) continue;
                if (MODE != 2 && ((claimed[idxB >> 5] >> (idxB & 31)) & 1u)) continue;    // :282 / :766
                const uint32_t d = hamming256(q, cd);
                if (MODE == 2) {
                    if (d > ORBM_TH_LOW) continue;                                        // :993 (first half)
                    if (!st1 && !cst2) {
                        const float dex = __fsub_rn(ex, cx2), dey = __fsub_rn(ey, cy2);
                        if (__fadd_rn(__fmul_rn(dex, dex), __fmul_rn(dey, dey)) < __fmul_rn(100.f, B.sf[coct2])) continue;  // :1007
                    }
                    if (!check_dist_epipolar_line(x1, y1, cx2, cy2, B.sigma2[coct2], F12)) continue;  // :1014
                    // "dist > bestDist -> skip" makes an equal distance REPLACE: lowest dist, LAST position wins
                    best = min(best, (d << 16) | (0xFFFFu - c));
                } else {
                    const uint32_t key = (d << 16) | c;
                    second = umed3(best, second, key);
                    best = min(best, key);
                }
            }
            // Nothing can be accepted unless some lane holds a distance within the threshold (:304 / :792 / :993):
            // one ballot saves the cross-lane top-2 reduction for every query without a plausible match.
            if (!__any((best >> 16) <= (uint32_t)(MODE == 1 ? ORBM_TH_LOW - 1 : ORBM_TH_LOW))) continue;
            if (MODE == 2) {
                best = wave_min(best);
                if (best != INITKEY) {                                                    // bestIdx2 >= 0  (:1021)
                    const uint32_t pos = 0xFFFFu - (best & 0xFFFFu);
                    const int idx2 = (int)B.fv_idx[offB + pos];
                    if (lane == 0) {
                        match[idxA] = idx2;
                        if (check_ori) atomicAdd(&hist[rot_bin(A.angle[idxA], B.angle[idx2])], 1);
                        my_matches++;
                    }
                }
            } else {
                wave_top2(best, second);
                const int bd = (int)(best >> 16), sd = (int)(second >> 16);
                const bool pass = MODE == 0 ? (bd <= ORBM_TH_LOW) : (bd < ORBM_TH_LOW);   // :304 / :792
                if (pass && (float)bd < __fmul_rn(ratio, (float)sd)) {                     // :308 / :796
                    const int idxBbest = (int)B.fv_idx[offB + (best & 0xFFFFu)];
                    if (lane == 0) {
                        atomicOr(&claimed[idxBbest >> 5], 1u << (idxBbest & 31));          // :312 / :799
                        if (MODE == 0) match[idxBbest] = idxA; else match[idxA] = idxBbest;
                        if (check_ori) atomicAdd(&hist[rot_bin(A.angle[idxA], angB[idxBbest])], 1);
                        my_matches++;
                    }
                    // the claim must be visible to this wave's next query (same node => same wave)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    }
    if (lane == 0 && my_matches) atomicAdd(&nmatches, my_matches);
    __syncthreads();
    if (SPLIT) {
        if (tid < ORBM_HISTO_LENGTH && hist[tid]) atomicAdd(&pair_acc[p * 32 + tid], hist[tid]);
        if (tid == 0 && nmatches) atomicAdd(&pair_acc[p * 32 + 30], nmatches);
        return;
    }
    if (check_ori) {
        if (tid == 0) {
            int i1, i2, i3;
            three_maxima(hist, i1, i2, i3);
            keep[0] = i1; keep[1] = i2; keep[2] = i3;
        }
        __syncthreads();
        const int k1 = keep[0], k2 = keep[1], k3 = keep[2];
        int removed = 0;
        for (int i = tid; i < nOut; i += 256) {
            const int m = match[i];
            if (m < 0) continue;
            const int idxA = MODE == 0 ? m : i, idxB = MODE == 0 ? i : m;
            const int bin = rot_bin(A.angle[idxA], angB[idxB]);
            if (bin != k1 && bin != k2 && bin != k3) { match[i] = -1; removed++; }        // :368-381 etc.
        }
        if (removed) atomicSub(&nmatches, removed);
        __syncthreads();
    }
    if (tid == 0) counts[p] = nmatches;
}

// ---- second form of the same searches: QUERIES on lanes --------------------------------------------------------------
// A vocabulary node holds ~20 features of each frame, so "64 lanes over one node's candidates" leaves two thirds of
// the lanes idle and pays a cross-lane reduction per query.  Here a wave packs the queries of several of its nodes
// into its 64 lanes (one query per lane) and every lane scans its own node's candidates serially: no reductions, the
// strict-'<' first-wins order is the loop order.  Claims only couple queries of the same node, which sit on consecutive
// lanes in query order: after the scan (which ignores claims) the lanes of a node are finalised one position at a
// time; a lane whose best or second-best candidate has been claimed meanwhile rescans with the claimed ones left out,
// which reproduces the sequential state exactly (claims only ever remove candidates).
template <int MODE, bool SPLIT>
__global__ __launch_bounds__(256) void bowq_kernel(const FrameDev *__restrict__ table, const int32_t *__restrict__ fa,
                                                   const int32_t *__restrict__ fb, float ratio, int check_ori,
                                                   int only_stereo, const float *__restrict__ f12all,
                                                   const float *__restrict__ epiall, int stride,
                                                   int32_t *__restrict__ match_all, int32_t *__restrict__ counts,
                                                   int splits, int *__restrict__ pair_acc) {
    __shared__ uint32_t claimed[MAX_N / 32];  // bit per b-side keypoint (a7: vpMapPointMatches set, a8: vbMatched2)
    __shared__ uint4 s_rows[4][2 * BOWQ_STAGE];   // per wave: the candidate rows of the nodes it is working on
    __shared__ uint32_t s_meta[4][BOWQ_STAGE];    // keypoint index | mp_flags << 16 of each staged candidate
    __shared__ int hist[ORBM_HISTO_LENGTH];
    __shared__ int keep[3];
    __shared__ int nmatches;
    const int p = SPLIT ? (int)blockIdx.x / splits : (int)xcd_remap(blockIdx.x, gridDim.x);
    const int part = SPLIT ? (int)blockIdx.x - p * splits : 0;
    const int nparts = SPLIT ? splits : 1;
    const FrameG A(table[fa[p]]), B(table[fb[p]]);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nOut = MODE == 0 ? B.n : A.n;
    int32_t *match = match_all + (size_t)p * stride;
    if (!SPLIT) for (int i = tid; i < nOut; i += 256) match[i] = -1;
    if (MODE != 2)
        for (int i = tid; i < (B.n + 31) / 32; i += 256) claimed[i] = 0;
    if (tid < ORBM_HISTO_LENGTH) hist[tid] = 0;
    if (tid == 0) nmatches = 0;
    __syncthreads();
    uint4 *w_rows = s_rows[wave];
    uint32_t *w_meta = s_meta[wave];

    const gptr<const float> angB = MODE == 0 ? B.angle_raw : B.angle;  // :323 reads F.mvKeys on the frame side
    float F12[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    float ex = 0.f, ey = 0.f;
    if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 9; i++) F12[i] = f12all[(size_t)p * 9 + i];
        ex = epiall[2 * p]; ey = epiall[2 * p + 1];
    }
    const uint32_t thr = (uint32_t)(MODE == 1 ? ORBM_TH_LOW - 1 : ORBM_TH_LOW);           // :304 / :792 / :993
    auto wave_max = [&](int v) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
        return __builtin_amdgcn_readfirstlane(v);
    };

    int my_matches = 0;
    const int first = part * 4 + wave, nstride = 4 * nparts;
    for (int kb = 0; first + kb * nstride < A.n_nodes; kb += 64) {
        // 64 of this wave's nodes, one per lane: the common-node lookup (lower_bound walk of :248-352) runs in parallel
        const int ia = first + (kb + lane) * nstride;
        uint32_t nA = 0, nB = 0, offA = 0, offB = 0;
        if (ia < A.n_nodes) {
            // two keyframes usually populate the same vocabulary nodes, so the node sits at the same index on both sides:
            // one probe there before the binary search (seven dependent loads)
            const uint32_t node_id = A.fv_node[ia];
            const int guess = min(ia, B.n_nodes - 1);
            const int ib = (guess >= 0 && B.fv_node[guess] == node_id) ? guess : find_node(B.fv_node, B.n_nodes, node_id);
            if (ib >= 0) {
                offA = A.fv_off[ia]; nA = A.fv_off[ia + 1] - offA;
                offB = B.fv_off[ib]; nB = B.fv_off[ib + 1] - offB;
                if (nB == 0) nA = 0;
            }
        }
        // Nodes are independent (a keypoint belongs to one node), so their order is free: sorted by candidate count,
        // the nodes packed into one group of 64 lanes have similar scan lengths (bitonic sort over the wave, descending).
        {
            uint32_t key = ((nA ? 0xFFFFu - min(nB, 0xFFFFu) : 0xFFFFu) << 6) | (uint32_t)lane;
#pragma unroll
            for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
                for (int j = k >> 1; j > 0; j >>= 1) {
                    const uint32_t other = (uint32_t)__shfl_xor((int)key, j, 64);
                    const bool asc = (lane & k) == 0, lower = (lane & j) == 0;
                    key = (lower == asc) ? min(key, other) : max(key, other);
                }
            const int from = (int)(key & 63u);
            nA = (uint32_t)__shfl((int)nA, from, 64); nB = (uint32_t)__shfl((int)nB, from, 64);
            offA = (uint32_t)__shfl((int)offA, from, 64); offB = (uint32_t)__shfl((int)offB, from, 64);
        }
        int node = 0;                 // next node (a lane of this batch) with queries left
        uint32_t node_done = 0;       // its queries already handled (nodes with more than 64 queries span groups)
        for (;;) {
            // ---- pack the next queries, node by node, into the 64 lanes
            int fill = 0, my_node = -1;
            uint32_t my_q = 0, my_take = 1, my_cbase = 0, csum = 0;
            int my_s = 0;
            while (node < 64 && fill < 64) {
                const uint32_t rem = (uint32_t)__builtin_amdgcn_readlane((int)nA, node) - node_done;
                if (rem == 0) { node++; node_done = 0; continue; }
                const uint32_t take = min(rem, (uint32_t)(64 - fill));
                if (lane >= fill && lane < fill + (int)take) {
                    my_node = node; my_s = lane - fill; my_q = node_done + (uint32_t)my_s; my_take = take; my_cbase = csum;
                }
                csum += (uint32_t)__builtin_amdgcn_readlane((int)nB, node);
                fill += (int)take;
                if (take == rem) { node++; node_done = 0; } else { node_done += take; break; }
            }
            if (fill == 0) break;
            const int src = max(my_node, 0);
            const uint32_t mOffA = (uint32_t)__shfl((int)offA, src, 64), mOffB = (uint32_t)__shfl((int)offB, src, 64);
            const uint32_t nb_src = (uint32_t)__shfl((int)nB, src, 64);   // all lanes take part: a shuffle cannot read a lane that sits it out
            const uint32_t mNB = my_node >= 0 ? nb_src : 0u;
            // ---- this lane's query
            bool qvalid = my_node >= 0;
            int idxA = 0;
            uint32_t q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            float x1 = 0.f, y1 = 0.f;
            bool st1 = false;
            if (qvalid) {
                idxA = (int)A.fv_idx[mOffA + my_q];
                const uint32_t fA = A.mp_flags[idxA];
                if (MODE == 2) {
                    st1 = A.uright[idxA] >= 0;
                    if ((fA & ORBM_MP_PRESENT) || (only_stereo && !st1)) qvalid = false;   // :932, :939-941
                } else {
                    if (!(fA & ORBM_MP_PRESENT) || (fA & ORBM_MP_BAD)) qvalid = false;      // :263-267 / :746-749
                }
                if (qvalid) {
                    load_row(A.desc, idxA, q);
                    if (MODE == 2) { x1 = A.x[idxA]; y1 = A.y[idxA]; }
                }
            }
            // ---- stage the candidate rows of the group's nodes in LDS: the lanes of a node share the work; every row is
            // then read from HBM/L2 once per group instead of once per lane (groups with more rows than fit read global)
            const bool staged = csum <= (uint32_t)BOWQ_STAGE;
            if (staged) {
                if (my_node >= 0)
                    for (uint32_t t = (uint32_t)my_s; t < mNB; t += my_take) {
                        const int idxB = (int)B.fv_idx[mOffB + t];
                        const gptr<const v4u> g = (gptr<const v4u>)B.desc + 2 * (size_t)idxB;
                        const v4u g0 = g[0], g1 = g[1];
                        w_rows[2 * (my_cbase + t)] = make_uint4(g0.x, g0.y, g0.z, g0.w);
                        w_rows[2 * (my_cbase + t) + 1] = make_uint4(g1.x, g1.y, g1.z, g1.w);
                        // the frame side's map-point flags play no part in SearchByBoW(KF, F) (:282 tests the match slot only): no second, dependent load
                        w_meta[my_cbase + t] = (uint32_t)idxB | (MODE == 0 ? 0u : (uint32_t)B.mp_flags[idxB] << 16);
                    }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            auto staged_candidate = [&](uint32_t t, int &idxB, uint32_t (&cd)[8], uint32_t &fl) {
                const uint4 lo = w_rows[2 * (my_cbase + t)], hi = w_rows[2 * (my_cbase + t) + 1];
                const uint32_t m = w_meta[my_cbase + t];
                cd[0] = lo.x; cd[1] = lo.y; cd[2] = lo.z; cd[3] = lo.w; cd[4] = hi.x; cd[5] = hi.y; cd[6] = hi.z; cd[7] = hi.w;
                idxB = (int)(m & 0xFFFFu); fl = m >> 16;
            };
            auto candidate_index = [&](uint32_t t) {
                return staged ? (int)(w_meta[my_cbase + t] & 0xFFFFu) : (int)B.fv_idx[mOffB + t];
            };
            // one candidate of this lane's node, already fetched: static gates + distance.  False when the reference skips it.
            auto evaluate = [&](int idxB, const uint32_t (&cd)[8], uint32_t fl, bool skip_claimed, uint32_t &d) {
                if (MODE == 1 && (!(fl & ORBM_MP_PRESENT) || (fl & ORBM_MP_BAD))) return false;    // :766-770
                if (MODE != 2 && skip_claimed && ((claimed[idxB >> 5] >> (idxB & 31)) & 1u)) return false;   // :282 / :766
                d = hamming256(q, cd);
                if (MODE == 2) {
                    if (d > ORBM_TH_LOW) return false;                                            // :993 (first half)
                    const bool cst2 = B.uright[idxB] >= 0;
                    if ((fl & ORBM_MP_PRESENT) || (only_stereo && !cst2)) return false;             // :970, :979-981
                    const float cx2 = B.x[idxB], cy2 = B.y[idxB];
                    const int coct2 = B.octave[idxB];
                    if (!st1 && !cst2) {
                        const float dex = __fsub_rn(ex, cx2), dey = __fsub_rn(ey, cy2);
                        if (__fadd_rn(__fmul_rn(dex, dex), __fmul_rn(dey, dey)) < __fmul_rn(100.f, B.sf[coct2])) return false;  // :1007
                    }
                    if (!check_dist_epipolar_line(x1, y1, cx2, cy2, B.sigma2[coct2], F12)) return false;  // :1014
                }
                return true;
            };
            auto update = [&](uint32_t d, uint32_t t, uint32_t &best, uint32_t &second) {
                if (MODE == 2) {
                    // "dist > bestDist -> skip" makes an equal distance REPLACE: lowest dist, LAST position wins
                    best = min(best, (d << 16) | (0xFFFFu - t));
                } else {
                    const uint32_t key = (d << 16) | t;
                    second = umed3(best, second, key);
                    best = min(best, key);
                }
            };
            // the first scan keeps the four smallest keys in order: when claims made meanwhile take candidates away, the
            // best two of what is left are the first two unclaimed of the four (anything outside the four is larger)
            auto update4 = [&](uint32_t d, uint32_t t, uint32_t &k0, uint32_t &k1, uint32_t &k2, uint32_t &k3) {
                uint32_t x = (d << 16) | t, lo;
                lo = min(k0, x); x = max(k0, x); k0 = lo;
                lo = min(k1, x); x = max(k1, x); k1 = lo;
                lo = min(k2, x); x = max(k2, x); k2 = lo;
                k3 = min(k3, x);
            };
            // rescan of one lane (rare): candidates claimed meanwhile are left out
            auto scan = [&](bool skip_claimed, uint32_t &best, uint32_t &second) {
                best = INITKEY; second = INITKEY;
                for (uint32_t t = 0; t < mNB; t++) {
                    int idxB; uint32_t cd[8], fl;
                    if (staged) staged_candidate(t, idxB, cd, fl);
                    else { idxB = (int)B.fv_idx[mOffB + t]; load_row(B.desc, idxB, cd); fl = MODE == 0 ? 0u : B.mp_flags[idxB]; }
                    uint32_t d = 0;
                    if (evaluate(idxB, cd, fl, skip_claimed, d)) update(d, t, best, second);
                }
            };
            // ---- scan: every lane against its own node's candidates, claims ignored.  The loads run ahead of the
            // arithmetic: index two candidates ahead, descriptor and flags one ahead (lanes of a node read the same rows).
            uint32_t best = INITKEY, second = INITKEY, third = INITKEY, fourth = INITKEY;
            auto note = [&](uint32_t d, uint32_t t) {
                if (MODE == 2) update(d, t, best, second); else update4(d, t, best, second, third, fourth);
            };
            if (staged && MODE != 2) {
                // no branches and no masked lanes inside: a lane past the end of its node (or without a query) reads its
                // last row again and inserts a key that sorts after everything; the next candidate's row is read from LDS
                // while this one is compared
                const int tmax = wave_max(qvalid ? (int)mNB : 0);
                const uint32_t tlast = mNB ? mNB - 1 : 0;
                const uint4 *rowp = w_rows + 2 * my_cbase;
                const uint32_t *metap = w_meta + my_cbase;
                auto step = [&](uint32_t t, const uint4 &lo, const uint4 &hi, uint32_t m) {
                    const uint32_t cd[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    bool ok = qvalid && t < mNB;
                    if (MODE == 1) ok = ok && ((m >> 16) & (ORBM_MP_PRESENT | ORBM_MP_BAD)) == ORBM_MP_PRESENT;   // :766-770
                    const uint32_t d = hamming256(q, cd);
                    uint32_t x = ok ? (d << 16) | t : 0xFFFFFFFFu, lo4;
                    lo4 = min(best, x); x = max(best, x); best = lo4;
                    lo4 = min(second, x); x = max(second, x); second = lo4;
                    lo4 = min(third, x); x = max(third, x); third = lo4;
                    fourth = min(fourth, x);
                };
                // two candidates per pass: the LDS reads of the second are in flight while the first is compared
                for (int t = 0; t < tmax; t += 2) {
                    const uint32_t t0 = min((uint32_t)t, tlast), t1 = min((uint32_t)t + 1u, tlast);
                    const uint4 lo0 = rowp[2 * t0], hi0 = rowp[2 * t0 + 1], lo1 = rowp[2 * t1], hi1 = rowp[2 * t1 + 1];
                    const uint32_t m0 = metap[t0], m1 = metap[t1];
                    step((uint32_t)t, lo0, hi0, m0);
                    step((uint32_t)t + 1u, lo1, hi1, m1);
                }
            } else if (staged) {
                const int tmax = wave_max(qvalid ? (int)mNB : 0);
                for (int t = 0; t < tmax; t++) {
                    if (qvalid && (uint32_t)t < mNB) {
                        int idxB; uint32_t cd[8], fl, d = 0;
                        staged_candidate((uint32_t)t, idxB, cd, fl);
                        if (evaluate(idxB, cd, fl, false, d)) note(d, (uint32_t)t);
                    }
                }
            } else {
                const int tmax = wave_max(qvalid ? (int)mNB : 0);
                const uint32_t last = mNB ? mNB - 1 : 0;
                const bool on = qvalid && mNB > 0;
                int idx_cur = on ? (int)B.fv_idx[mOffB] : 0;
                int idx_nxt = on ? (int)B.fv_idx[mOffB + min(1u, last)] : 0;
                uint32_t row_nxt[8];
                load_row(B.desc, idx_cur, row_nxt);
                uint32_t fl_nxt = MODE == 0 ? 0u : B.mp_flags[idx_cur];
                for (int t = 0; t < tmax; t++) {
                    uint32_t cd[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) cd[i] = row_nxt[i];
                    const uint32_t fl = fl_nxt;
                    const int idxB = idx_cur;
                    // issue the loads of the next candidates before touching this one
                    idx_cur = idx_nxt;
                    load_row(B.desc, idx_cur, row_nxt);
                    fl_nxt = MODE == 0 ? 0u : B.mp_flags[idx_cur];
                    idx_nxt = on ? (int)B.fv_idx[mOffB + min((uint32_t)t + 2u, last)] : 0;
                    if (on && (uint32_t)t < mNB) {
                        uint32_t d = 0;
                        if (evaluate(idxB, cd, fl, false, d)) note(d, (uint32_t)t);
                    }
                }
            }
            if (MODE == 2) {
                if (qvalid && best != INITKEY) {                                                  // bestIdx2 >= 0  (:1021)
                    const int idx2 = candidate_index(0xFFFFu - (best & 0xFFFFu));
                    match[idxA] = idx2;
                    if (check_ori) atomicAdd(&hist[rot_bin(A.angle[idxA], B.angle[idx2])], 1);
                    my_matches++;
                }
                continue;
            }
            // ---- finalise in query order inside each node.  A best distance above the threshold can only get worse.
            if ((best >> 16) > thr) qvalid = false;
            // Queries still in play are finalised in order inside each node, the nodes of the group side by side: pass r takes
            // the r-th such query of every node (its rank among the node's lanes that still hold a query), so the group
            // needs as many passes as its fullest node has such queries, not one per occupied position.
            const unsigned long long vmask = __ballot(qvalid);
            const int node_first = lane - my_s;                               // first lane of this lane's node in the group
            const unsigned long long before = vmask & ((1ull << lane) - 1ull) & ~((1ull << max(node_first, 0)) - 1ull);
            const int my_rank = __popcll(before);
            const int npass = wave_max(qvalid ? my_rank + 1 : 0);
            for (int s2 = 0; s2 < npass; s2++) {
                if (qvalid && my_rank == s2) {
                    int ib1 = candidate_index(best & 0xFFFFu);
                    bool stale = (claimed[ib1 >> 5] >> (ib1 & 31)) & 1u;
                    if (!stale && second != INITKEY) {
                        const int ib2 = candidate_index(second & 0xFFFFu);
                        stale = (claimed[ib2 >> 5] >> (ib2 & 31)) & 1u;
                    }
                    if (stale) {
                        const uint32_t kk[4] = {best, second, third, fourth};
                        uint32_t left[2] = {INITKEY, INITKEY};
                        int nleft = 0;
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            bool gone = false;
                            if (kk[i] != INITKEY) {                       // INITKEY: the node has no further candidate
                                const int ib = candidate_index(kk[i] & 0xFFFFu);
                                gone = (claimed[ib >> 5] >> (ib & 31)) & 1u;
                            }
                            if (!gone && nleft < 2) { if (nleft == 0) left[0] = kk[i]; else left[1] = kk[i]; nleft++; }
                        }
                        if (nleft == 2) { best = left[0]; second = left[1]; }
                        else scan(true, best, second);                    // three of the four claimed: look again (rare)
                        if (best != INITKEY) ib1 = candidate_index(best & 0xFFFFu);
                    }
                    const int bd = (int)(best >> 16), sd = (int)(second >> 16);
                    const bool pass = MODE == 0 ? (bd <= ORBM_TH_LOW) : (bd < ORBM_TH_LOW);       // :304 / :792
                    if (pass && (float)bd < __fmul_rn(ratio, (float)sd)) {                         // :308 / :796
                        atomicOr(&claimed[ib1 >> 5], 1u << (ib1 & 31));                            // :312 / :799
                        if (MODE == 0) match[ib1] = idxA; else match[idxA] = ib1;
                        if (check_ori) atomicAdd(&hist[rot_bin(A.angle[idxA], angB[ib1])], 1);
                        my_matches++;
                    }
                }
                // claims must be visible to the later positions of the same node (same wave)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    if (my_matches) atomicAdd(&nmatches, my_matches);
    __syncthreads();
    if (SPLIT) {
        if (tid < ORBM_HISTO_LENGTH && hist[tid]) atomicAdd(&pair_acc[p * 32 + tid], hist[tid]);
        if (tid == 0 && nmatches) atomicAdd(&pair_acc[p * 32 + 30], nmatches);
        return;
    }
    if (check_ori) {
        if (tid == 0) {
            int i1, i2, i3;
            three_maxima(hist, i1, i2, i3);
            keep[0] = i1; keep[1] = i2; keep[2] = i3;
        }
        __syncthreads();
        const int k1 = keep[0], k2 = keep[1], k3 = keep[2];
        int removed = 0;
        for (int i = tid; i < nOut; i += 256) {
            const int m = match[i];
            if (m < 0) continue;
            const int idxA = MODE == 0 ? m : i, idxB = MODE == 0 ? i : m;
            const int bin = rot_bin(A.angle[idxA], angB[idxB]);
            if (bin != k1 && bin != k2 && bin != k3) { match[i] = -1; removed++; }        // :368-381 etc.
        }
        if (removed) atomicSub(&nmatches, removed);
        __syncthreads();
    }
    if (tid == 0) counts[p] = nmatches;
}

// rotation-consistency filter + count of a pair whose nodes were matched by several workgroups (:355-383 / :827-849 / :1062-1082)
template <int MODE>
__global__ __launch_bounds__(256) void bow_finish_kernel(const FrameDev *__restrict__ table, const int32_t *__restrict__ fa,
                                                         const int32_t *__restrict__ fb, int check_ori, int stride,
                                                         int32_t *__restrict__ match_all, int32_t *__restrict__ counts,
                                                         const int *__restrict__ pair_acc, int encoded) {
    // encoded (the matrix-core form with the orientation check on): a match was written as index | rotation bin << 16, so this
    // pass needs neither frame's angles: it keeps or drops by the bin and leaves the plain index
    __shared__ int hist[ORBM_HISTO_LENGTH];
    __shared__ int keep[3];
    __shared__ int removed_all;
    const int p = blockIdx.x, tid = threadIdx.x;
    const FrameG A(table[fa[p]]), B(table[fb[p]]);
    const int nOut = MODE == 0 ? B.n : A.n;
    const gptr<const float> angB = MODE == 0 ? B.angle_raw : B.angle;
    int32_t *match = match_all + (size_t)p * stride;
    if (tid < ORBM_HISTO_LENGTH) hist[tid] = pair_acc[p * 32 + tid];
    if (tid == 0) removed_all = 0;
    __syncthreads();
    if (check_ori) {
        if (tid == 0) {
            int i1, i2, i3;
            three_maxima(hist, i1, i2, i3);
            keep[0] = i1; keep[1] = i2; keep[2] = i3;
        }
        __syncthreads();
        const int k1 = keep[0], k2 = keep[1], k3 = keep[2];
        int removed = 0;
        if (encoded) {
            for (int i = tid; i < nOut; i += 256) {
                const int m = match[i];
                if (m < 0) continue;
                const int bin = m >> 16;
                if (bin != k1 && bin != k2 && bin != k3) { match[i] = -1; removed++; }
                else match[i] = m & 0xFFFF;
            }
        } else {
            for (int i = tid; i < nOut; i += 256) {
                const int m = match[i];
                if (m < 0) continue;
                const int idxA = MODE == 0 ? m : i, idxB = MODE == 0 ? i : m;
                const int bin = rot_bin(A.angle[idxA], angB[idxB]);
                if (bin != k1 && bin != k2 && bin != k3) { match[i] = -1; removed++; }
            }
        }
        if (removed) atomicAdd(&removed_all, removed);
        __syncthreads();
    }
    if (tid == 0) counts[p] = pair_acc[p * 32 + 30] - removed_all;
}

// the split forms start from match rows of -1 and zeroed per-pair accumulators: one launch for both (two memsets are two
// launches, and a small batch is a handful of launches in all)
__global__ __launch_bounds__(256) void bow_init_kernel(int32_t *__restrict__ match, size_t n_match, int *__restrict__ acc, int n_acc) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, nthreads = (size_t)gridDim.x * 256;
    for (size_t k = i; k < n_match; k += nthreads) match[k] = -1;
    for (size_t k = i; k < (size_t)n_acc; k += nthreads) acc[k] = 0;
}
void launch_bow_init(orbm_ctx *ctx, int32_t *match, size_t n_match, int *acc, int n_acc) {
    const size_t most = std::max(n_match, (size_t)n_acc);
    const unsigned grid = (unsigned)std::max<size_t>(1, std::min<size_t>(4096, (most + 1023) / 1024));
    hipLaunchKernelGGL(bow_init_kernel, dim3(grid), dim3(256), 0, ctx->stream, match, n_match, acc, n_acc);
}

// tuning / test knob: 0 = auto (batches of g_bow_split_below pairs and more: the matrix-core form of bowm.hip for a7 / a8,
// bowq_kernel for a9; smaller batches share a pair between several workgroups: queries on lanes (bowq_kernel) above 96
// pairs, candidates on lanes (bow_kernel) below), 1 = bow_kernel always, 2 = bowq_kernel always, 3 = bowq_kernel with one
// workgroup per pair whatever the batch size, 5 = the matrix-core form whatever the batch size
int g_bow_variant = 0;
int g_bow_split_below = 2048;   // batches of fewer pairs share each pair between several workgroups
int g_bowm_wgs = 4096;          // matrix-core form: workgroups a small batch is spread over
constexpr int BOWM_SPLIT_BELOW = 900;   // ... below this many pairs (scripts/bow64_perf.py: 512 pairs 0.038 ms split / 0.041 whole, 1024 pairs 0.053 / 0.051)

// Allocation-free variant (orbm_search_bow_batch_noalloc): the per-pair accumulators of the split path (30 histogram bins +
// the match count, 32 ints per pair) live in this file-scope device array instead of the context's grow-only workspace.
// Every GPU the library runs on holds its own copy (256 KB) until the process ends; the host reaches it only through the
// address hipGetSymbolAddress returns for that device (the array's name is a host-side stand-in, not a device address).
constexpr int NOALLOC_MAX_PAIRS = 2048;
static __device__ __align__(256) int scratch[NOALLOC_MAX_PAIRS * 32];
constexpr int NOALLOC_MAX_DEVICES = 64;
int *g_scratch_ptr[NOALLOC_MAX_DEVICES];   // per device; written once under g_scratch_mutex, read after orbm_create returned
std::mutex g_scratch_mutex;

int launch_bow(orbm_ctx *ctx, int mode, int n_pairs, const int32_t *frame_a, const int32_t *frame_b, float ratio,
               int check_ori, int only_stereo, const float *f12, const float *epipole, int32_t stride,
               int32_t *match, int32_t *counts, int io_on_device, const char *who, int *static_acc = nullptr) {
    if (!ctx) return ORBM_EINVAL;
    if (n_pairs < 0 || (n_pairs > 0 && (!frame_a || !frame_b || !match || !counts)) || stride < 0 ||
        (mode == 2 && n_pairs > 0 && (!f12 || !epipole)))
        return orbm_fail(ctx, ORBM_EINVAL, "%s: null argument or negative size", who);
    if (n_pairs == 0) return ORBM_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!io_on_device) {
        int rc = orbm_check_frames(ctx, n_pairs, frame_a, "frame_a");
        if (!rc) rc = orbm_check_frames(ctx, n_pairs, frame_b, "frame_b");
        if (rc) return rc;
        for (int p = 0; p < n_pairs; p++) {
            const int need = mode == 0 ? ctx->h_frames[frame_b[p]].n : ctx->h_frames[frame_a[p]].n;
            if (need > stride) return orbm_fail(ctx, ORBM_EINVAL, "%s: stride %d < %d rows needed by pair %d", who, stride, need, p);
        }
    }
    const size_t out_elems = (size_t)n_pairs * stride;
    // a7 / a8 take the matrix-core form (bowm.hip) when the ratio leaves few distances relevant (not the allocation-free entry
    // point: the form reads a FeatureVector-ordered copy of the frames' descriptors, built here on first use -- before
    // anything is staged, the build uses the same staging buffers)
    const int capd = mode == 2 ? ORBM_TH_LOW + 1 : orbm_bow_cap(ratio);   // a9 has no ratio test: relevant = d <= TH_LOW
    const bool one_wg_per_pair = !static_acc && (n_pairs >= g_bow_split_below || g_bow_variant == 3);
    const bool use_mfma = !static_acc && capd <= 128 && (g_bow_variant == 0 || g_bow_variant == 5);
    if (use_mfma) {
        int rc = io_on_device ? orbm_ensure_image(ctx, 5, 0, nullptr) : orbm_ensure_image(ctx, 5, n_pairs, frame_a);
        if (!rc && !io_on_device) rc = orbm_ensure_image(ctx, 5, n_pairs, frame_b);
        if (rc) return rc;
    }
    IoStage io(ctx, io_on_device);
    if (!io_on_device) {
        int rc = orbm_ws_reserve(ctx, (size_t)n_pairs * (12 + 44) + out_elems * 4 + 8192);
        if (rc) return rc;
    }
    const void *dfa, *dfb, *df12 = nullptr, *dep = nullptr; void *dm, *dc;
    int rc;
    if ((rc = io.in(frame_a, (size_t)n_pairs * 4, &dfa)) || (rc = io.in(frame_b, (size_t)n_pairs * 4, &dfb)) ||
        (rc = io.in(f12, (size_t)n_pairs * 36, &df12)) || (rc = io.in(epipole, (size_t)n_pairs * 8, &dep)) ||
        (rc = io.out(match, out_elems * 4, &dm)) || (rc = io.out(counts, (size_t)n_pairs * 4, &dc)))
        return rc;
    // One workgroup per pair needs a few thousand pairs to fill the chip (a pair keeps its workgroup 0.1-0.2 ms).  Below that
    // several workgroups share a pair's nodes, about 4096 workgroups in all and at most 32 (16) per pair; up to 96 pairs the
    // candidates-on-lanes kernel is the faster of the two forms, above the queries-on-lanes kernel (scripts/bow64_perf.py:
    // 64 / 256 / 512 / 1024 pairs 0.050 / 0.078 / 0.114 / 0.167 ms against 0.056 / 0.130 / 0.181 / 0.203 ms with one policy for all).
    const bool split_queries_on_lanes = g_bow_variant == 2 || (g_bow_variant != 1 && n_pairs > 96);
    // the matrix-core form keeps a workgroup about 50 us on a pair (a wave walks 25 nodes one after the other): below
    // g_bowm_wgs workgroups in all a pair's nodes are dealt to several workgroups (at most 16)
    const int splits = use_mfma ? (n_pairs >= std::min(g_bow_split_below, BOWM_SPLIT_BELOW) ? 1 : std::max(1, std::min(16, g_bowm_wgs / n_pairs)))
                       : one_wg_per_pair ? 1
                       : std::max(2, std::min(split_queries_on_lanes ? 16 : 32, (4096 + n_pairs - 1) / n_pairs));
    dim3 block(256);
    // (the kernels with one workgroup per pair set their own match rows; in host mode rows beyond a pair's own length are
    // part of what is copied back, so the whole buffer is set there)
    if (splits == 1 && !io_on_device && out_elems) HIPCHK(ctx, hipMemsetAsync(dm, 0xFF, out_elems * 4, ctx->stream));
    if (use_mfma) {
        int *acc = nullptr;
        if (splits > 1) {
            rc = orbm_ws2_reserve(ctx, (size_t)n_pairs * 32 * sizeof(int));
            if (rc) return rc;
            acc = (int *)ctx->ws2;
            launch_bow_init(ctx, (int32_t *)dm, out_elems, acc, n_pairs * 32);
        }
        rc = orbm_launch_bowm(ctx, mode, n_pairs, (const int32_t *)dfa, (const int32_t *)dfb, ratio, capd, check_ori, stride, (int32_t *)dm,
                              (int32_t *)dc, splits, acc, only_stereo, (const float *)df12, (const float *)dep);
        if (rc) return rc;
        if (splits > 1) {
            dim3 fgrid(n_pairs);
            if (mode == 0)
                hipLaunchKernelGGL((bow_finish_kernel<0>), fgrid, block, 0, ctx->stream, ctx->d_frames, (const int32_t *)dfa, (const int32_t *)dfb,
                                   check_ori, stride, (int32_t *)dm, (int32_t *)dc, (const int *)acc, check_ori ? 1 : 0);
            else if (mode == 2)
                hipLaunchKernelGGL((bow_finish_kernel<2>), fgrid, block, 0, ctx->stream, ctx->d_frames, (const int32_t *)dfa, (const int32_t *)dfb,
                                   check_ori, stride, (int32_t *)dm, (int32_t *)dc, (const int *)acc, check_ori ? 1 : 0);
            else
                hipLaunchKernelGGL((bow_finish_kernel<1>), fgrid, block, 0, ctx->stream, ctx->d_frames, (const int32_t *)dfa, (const int32_t *)dfb,
                                   check_ori, stride, (int32_t *)dm, (int32_t *)dc, (const int *)acc, check_ori ? 1 : 0);
        }
    } else if (splits == 1) {
        dim3 grid(n_pairs);
#define LAUNCH(M)                                                                                              \
    hipLaunchKernelGGL((g_bow_variant == 1 ? bow_kernel<M, false> : bowq_kernel<M, false>), grid, block, 0, ctx->stream, ctx->d_frames, (const int32_t *)dfa, \
                       (const int32_t *)dfb, ratio, check_ori, only_stereo, (const float *)df12, (const float *)dep, \
                       stride, (int32_t *)dm, (int32_t *)dc, 1, (int *)nullptr)
        if (mode == 0) LAUNCH(0); else if (mode == 1) LAUNCH(1); else LAUNCH(2);
#undef LAUNCH
    } else {
        if (!static_acc) {
            rc = orbm_ws2_reserve(ctx, (size_t)n_pairs * 32 * sizeof(int));
            if (rc) return rc;
        }
        int *acc = static_acc ? static_acc : (int *)ctx->ws2;
        launch_bow_init(ctx, (int32_t *)dm, out_elems, acc, n_pairs * 32);
        dim3 grid(n_pairs * splits), fgrid(n_pairs);
#define LAUNCH(M)                                                                                              \
    do {                                                                                                       \
        hipLaunchKernelGGL((split_queries_on_lanes ? bowq_kernel<M, true> : bow_kernel<M, true>), grid, block, 0, ctx->stream, ctx->d_frames, (const int32_t *)dfa, \
                           (const int32_t *)dfb, ratio, check_ori, only_stereo, (const float *)df12, (const float *)dep, \
                           stride, (int32_t *)dm, (int32_t *)dc, splits, acc);                                  \
        hipLaunchKernelGGL((bow_finish_kernel<M>), fgrid, block, 0, ctx->stream, ctx->d_frames, (const int32_t *)dfa,   \
                           (const int32_t *)dfb, check_ori, stride, (int32_t *)dm, (int32_t *)dc, (const int *)acc, 0);   \
    } while (0)
        if (mode == 0) LAUNCH(0); else if (mode == 1) LAUNCH(1); else LAUNCH(2);
#undef LAUNCH
    }
    HIPCHK(ctx, hipGetLastError());
    return io.finish();
}

}  // namespace

// Called by orbm_create with the context's device current: resolves that device's copy of `scratch` once.
int orbm_bow_scratch_resolve(orbm_ctx *ctx) {
    if (ctx->device < 0 || ctx->device >= NOALLOC_MAX_DEVICES) return ORBM_OK;   // such a device simply has no allocation-free entry
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[ctx->device]) return ORBM_OK;
    int *ptr = nullptr;
    HIPCHK(ctx, hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
    g_scratch_ptr[ctx->device] = ptr;
    return ORBM_OK;
}

// tuning / test knob (not part of the public ABI): see g_bow_variant
extern "C" void orbm__bow_variant(int v) { g_bow_variant = v; }
extern "C" void orbm__bow_split_below(int n_pairs) { g_bow_split_below = n_pairs > 0 ? n_pairs : 2048; }
extern "C" void orbm__bowm_wgs(int n) { g_bowm_wgs = n > 0 ? n : 4096; }

extern "C" int orbm_search_bow_batch(orbm_ctx *ctx, int variant, int n_pairs, const int32_t *frame_a,
                                     const int32_t *frame_b, float ratio, int check_ori, int32_t stride,
                                     int32_t *match, int32_t *counts, int io_on_device) {
    if (ctx && variant != 0 && variant != 1) return orbm_fail(ctx, ORBM_EINVAL, "search_bow_batch: variant must be 0 or 1");
    return launch_bow(ctx, variant, n_pairs, frame_a, frame_b, ratio, check_ori, 0, nullptr, nullptr, stride, match,
                      counts, io_on_device, "search_bow_batch");
}

extern "C" int orbm_search_bow_batch_noalloc(orbm_ctx *ctx, int variant, int n_pairs, const int32_t *frame_a,
                                             const int32_t *frame_b, float ratio, int check_ori, int32_t stride,
                                             int32_t *match, int32_t *counts) {
    if (!ctx) return ORBM_EINVAL;
    if (variant != 0 && variant != 1) return orbm_fail(ctx, ORBM_EINVAL, "search_bow_batch_noalloc: variant must be 0 or 1");
    if (n_pairs > NOALLOC_MAX_PAIRS)
        return orbm_fail(ctx, ORBM_ELIMIT, "search_bow_batch_noalloc: %d pairs exceed the static scratch buffer (%d pairs)", n_pairs,
                         NOALLOC_MAX_PAIRS);
    int *acc = ctx->device >= 0 && ctx->device < NOALLOC_MAX_DEVICES ? g_scratch_ptr[ctx->device] : nullptr;
    if (!acc) return orbm_fail(ctx, ORBM_EHIP, "search_bow_batch_noalloc: no static scratch buffer on device %d", ctx->device);
    return launch_bow(ctx, variant, n_pairs, frame_a, frame_b, ratio, check_ori, 0, nullptr, nullptr, stride, match, counts,
                      /*io_on_device=*/1, "search_bow_batch_noalloc", acc);
}

extern "C" int orbm_search_triang_batch(orbm_ctx *ctx, int n_pairs, const int32_t *frame_a, const int32_t *frame_b,
                                        const float *f12, const float *epipole, int only_stereo, int check_ori,
                                        int32_t stride, int32_t *match12, int32_t *counts, int io_on_device) {
    return launch_bow(ctx, 2, n_pairs, frame_a, frame_b, 1.0f, check_ori, only_stereo, f12, epipole, stride, match12,
                      counts, io_on_device, "search_triang_batch");
}
