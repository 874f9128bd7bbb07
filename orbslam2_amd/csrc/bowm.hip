// bowm.hip -- SearchByBoW (a7: ORBmatcher.cc:218-385, a8: :703-854) for LARGE batches of keyframe pairs, distances on the
// matrix cores.
//
// What the reference computes per common vocabulary node is a small dense block: every query of the node against every
// candidate of the node (about 20 x 20 for 2000 features over 100 nodes).  With the descriptor bits as +-1 (FP4 E2M1, exact)
// a 32 x 32 block of Hamming distances is four v_mfma_scale_f32_32x32x64_f8f6f4 (d = (256 - dot) / 2, see bf2nn.hip).  A
// node's rows are CONSECUTIVE in the FeatureVector-ordered copy of the descriptors (FrameDev::descfv, 32 bytes per row), so
// an operand is ONE coalesced 16-byte load per lane; the bits become FP4 values in registers (7 vector instructions per 32
// bits, expand_fp4) -- the pre-expanded image the brute-force search uses would move four times the bytes, and this path
// has no operand reuse to pay for them.  No LDS staging, no per-candidate loop.
//
// What the reference DECIDES per query depends on very few of those distances.  It accepts a query's best candidate iff
// bestDist1 <= TH_LOW (a7; '<' in a8) and (float)bestDist1 < ratio * (float)bestDist2 (:304-308 / :792-796).  Let capd be the
// smallest distance with ratio * (float)capd > TH_LOW: any second-best distance >= capd passes the ratio test for every
// acceptable best distance, exactly as "no second candidate" (256) does.  So only candidates with d < capd ("relevant":
// 67 for ratio 0.75, 84 for 0.6) can change a result, and between descriptors of different landmarks they are rare.  The
// wave folds the 16 dots a lane holds with v_max3 against that one threshold; only when some lane passes does it look at
// the values and append (query position, distance, candidate position) records to a short list in LDS.  The claims
// (vpMapPointMatches[idxF] / vbMatched2[idx2], which only couple queries of the SAME node because a keypoint belongs to
// one node) are then settled on the list alone:
//   * the usual case, no two records share a query or a candidate: every record decides for itself (second best = none);
//   * otherwise the records are taken query by query in the reference's order (ascending position in the node), best and
//     second best among the query's records whose candidate is still free, a claim removes the candidate for later queries;
//   * a node with more records than lanes, or too large for the record format, falls back to the generic loop (queries in
//     order, 64 lanes over the candidates, claimed bits in LDS): the same code path as bow_kernel in bow.hip.
// Rotation histogram, ComputeThreeMaxima and the erase pass as in bow.hip.  One 256-thread workgroup per pair, a wave takes
// every fourth node of the a-side FeatureVector.
#include <algorithm>
#include "orbm_internal.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr uint32_t INITKEY = (256u << 16) | 0xFFFFu;
constexpr int MAX_N = 65536;
constexpr int BM_LIST = 320;          // records a wave collects before it settles them (a node with more than 64 takes the generic loop)
constexpr uint32_t REC_MAX_Q = 4096, REC_MAX_T = 8192;   // record = q << 20 | d << 13 | t
// waves per SIMD: five (96 registers).  Six (80) were better while the tile loop waited on loads (round 3's first half: + 6 %);
// the loop of the tile list is bound by vector issue, six waves spill into it, and five are 3-5 % faster on pairs with matches
#define BOWM_WAVES 5
constexpr int TL_CAP = 128;           // tiles a wave lists before it multiplies them (a batch of 64 nodes with more takes several passes)
constexpr int BIN_ROWS = 4096;        // match rows up to this length keep their matches' rotation bins in LDS for the filter pass
#ifdef ORBM_DIAG
__device__ int d_bowm_diag;   // timing only, WRONG results: 1 = records are dropped before they are settled, 2 = no records are made, 4 = accepted matches are not applied
#define BOWM_DIAG(bit) (diag_bits & (bit))
__device__ unsigned long long d_bowm_trace[64 * 8];   // wall-clock stamps of every 128th workgroup's wave 0: start, set-up done, nodes looked up, tiles done, records settled, end
#define BOWM_STAMP(k) do { if ((blockIdx.x & 127) == 0 && blockIdx.x < 64 * 128 && tid == 0) d_bowm_trace[(blockIdx.x >> 7) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define BOWM_DIAG(bit) false
#define BOWM_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ void load_row(gptr<const uint32_t> desc, int idx, uint32_t (&r)[8]) {
    const gptr<const v4u> p = (gptr<const v4u>)desc + 2 * (size_t)idx;
    const v4u lo = p[0], hi = p[1];
    r[0] = lo.x; r[1] = lo.y; r[2] = lo.z; r[3] = lo.w; r[4] = hi.x; r[5] = hi.y; r[6] = hi.z; r[7] = hi.w;
}

__device__ __forceinline__ void wave_top2(uint32_t &best, uint32_t &second) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t ob = __shfl_xor(best, off, 64), os = __shfl_xor(second, off, 64);
        const uint32_t nb = min(best, ob);
        second = min(max(best, ob), min(second, os));
        best = nb;
    }
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, off, 64));
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

// 32 descriptor bits -> 32 FP4 values (+1.0 = 0x2 for a 0 bit, -1.0 = 0xA for a 1 bit), eight per dword: dword k takes bits
// 3 - k, 7 - k, ..., 31 - k, shifted up to the sign position of their nibble (3 shifts + 4 v_and_or_b32 per 32 bits).
// Here for TWO dwords that sit in a register pair: one v_lshlrev_b64 shifts both (4.3 cycles per wave against 4.1 for each
// v_lshlrev_b32, scripts/ubench_valu5.hip).  The bits that cross from the low into the high dword land at positions 0 .. 2,
// which the sign mask drops.  12 + 32 instead of 24 + 32 vector instructions per tile.
template <int S>
__device__ __forceinline__ uint64_t shl64(uint64_t x) {
    uint64_t r;
    asm("v_lshlrev_b64 %0, %1, %2" : "=v"(r) : "n"(S), "v"(x));
    return r;
}
// The last step is written in C on purpose: (x & sign) | ones is ONE v_and_or_b32 either way, but an asm statement is not a
// vector instruction to hipcc, and a matrix instruction needs two wait states after a vector write of one of its operands
// (no hardware interlock: with one, the first passes of the product read the register's OLD contents -- seen as missing
// matches in rows 0-11 of a tile).  hipcc inserts them for instructions it knows.
__device__ __forceinline__ uint32_t and_or_c(uint32_t x, uint32_t bits) { return (x & 0x88888888u) | bits; }
__device__ __forceinline__ void expand_fp4x2(uint64_t w, uint32_t ones, v8i &f0, v8i &f1) {
    const uint64_t t3 = shl64<3>(w), t2 = shl64<2>(w), t1 = shl64<1>(w);
    f0 = (v8i){(int)and_or_c((uint32_t)t3, ones), (int)and_or_c((uint32_t)t2, ones), (int)and_or_c((uint32_t)t1, ones), (int)and_or_c((uint32_t)w, ones), 0, 0, 0, 0};
    f1 = (v8i){(int)and_or_c((uint32_t)(t3 >> 32), ones), (int)and_or_c((uint32_t)(t2 >> 32), ones), (int)and_or_c((uint32_t)(t1 >> 32), ones), (int)and_or_c((uint32_t)(w >> 32), ones), 0, 0, 0, 0};
}

// a wave-uniform address, said so that the compiler keeps it in scalar registers (and the per-lane part a 32-bit offset of the load)
__device__ __forceinline__ gptr<const char> uniform_ptr(gptr<const char> p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (gptr<const char>)((uint64_t)lo | ((uint64_t)hi << 32));
}

// MODE 0: a7 (KF -> F), 1: a8 (KF -> KF), 2: a9 (SearchForTriangulation, ORBmatcher.cc:868-1093: no ratio test and no claims --
// a query takes the candidate of least distance <= TH_LOW among those that pass the stereo / epipole / epipolar-line
// gates, the LAST one among equals; relevant = d <= TH_LOW, the gates are evaluated per record)
// SPLIT = false: one workgroup per pair does everything.  SPLIT = true: `splits` workgroups share a pair's nodes (small
// batches: a single SearchByBoW call would otherwise run on one CU); the match rows were pre-set to -1 by the caller, histogram
// and count go to pair_acc (32 ints per pair) and bow_finish_kernel (bow.hip) applies the rotation filter afterwards.
template <int MODE, bool SPLIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BOWM_WAVES, BOWM_WAVES))) void bowm_kernel(const FrameDev *__restrict__ table, const int32_t *__restrict__ fa,
                                                   const int32_t *__restrict__ fb, float ratio, int capd, int check_ori, int stride,
                                                   int32_t *__restrict__ match_all, int32_t *__restrict__ counts, int splits,
                                                   int *__restrict__ pair_acc, int only_stereo, const float *__restrict__ f12all,
                                                   const float *__restrict__ epiall) {
    __shared__ uint32_t claimed[MAX_N / 32];   // generic loop only: bit per b-side keypoint (a7: vpMapPointMatches set, a8: vbMatched2)
    __shared__ uint32_t s_list[4][BM_LIST];
    __shared__ uint8_t s_tags[4][BM_LIST];      // which of the wave's current 64 nodes a record belongs to
    __shared__ uint32_t s_cand[4];              // candidates named by the records of the tile at hand, one bit each (zero between tiles)
    __shared__ uint2 s_tiles[4][TL_CAP];        // the wave's tile list: two words per 32 x 32 tile (see the tile loop)
    __shared__ __align__(16) uint8_t s_bin[SPLIT ? 16 : BIN_ROWS];   // rotation bin of the match in row i of the pair's match row, 0xFF = none (rows < BIN_ROWS)
    __shared__ int hist[ORBM_HISTO_LENGTH];
    __shared__ int keep[3];
    __shared__ int nmatches;
    const int p = SPLIT ? (int)blockIdx.x / splits : (int)xcd_remap(blockIdx.x, gridDim.x);
    const int part = SPLIT ? (int)blockIdx.x - p * splits : 0, nparts = SPLIT ? splits : 1;
    const FrameG A(table[fa[p]]), B(table[fb[p]]);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, h = lane >> 5;
#ifdef ORBM_DIAG
    const int diag_bits = d_bowm_diag;
#endif
    BOWM_STAMP(0);
    const int nOut = MODE == 0 ? B.n : A.n;
    int32_t *match = match_all + (size_t)p * stride;
    if (!SPLIT) for (int i = tid; i < nOut; i += 256) match[i] = -1;
    const bool bins_in_lds = !SPLIT && check_ori && nOut <= BIN_ROWS;
    if (bins_in_lds) for (int i = tid * 16; i < nOut; i += 256 * 16) *reinterpret_cast<v4u *>(&s_bin[i]) = v4u{~0u, ~0u, ~0u, ~0u};
    for (int i = tid; i < (B.n + 31) / 32; i += 256) claimed[i] = 0;
    if (tid < ORBM_HISTO_LENGTH) hist[tid] = 0;
    if (tid == 0) nmatches = 0;
    if (tid < 4) s_cand[tid] = 0;
    __syncthreads();
    BOWM_STAMP(1);

    const gptr<const float> angB = MODE == 0 ? B.angle_raw : B.angle;  // :323 reads F.mvKeys on the frame side
    const int thr = MODE == 1 ? ORBM_TH_LOW - 1 : ORBM_TH_LOW;          // :304 / :792
    const float Tf = (float)(256 - 2 * capd);                          // dot > Tf  <=>  distance < capd
    const int Ti = __float_as_int(Tf);                                 // capd <= 128: Tf >= 0, and a float >= 0 orders like its bits
    uint32_t ones = 0x22222222u;                                       // +1.0 in every nibble, kept in a vector register
    asm volatile("" : "+v"(ones));
    const gptr<const v4u> rowsA = (gptr<const v4u>)A.descfv, rowsB = (gptr<const v4u>)B.descfv;   // FeatureVector order
    const uint32_t lane_off = (uint32_t)(32 * col + 16 * h);
    // (LDS pointers with their address space spelled out: 32 bits each, where `uint32_t *` is a 64-bit generic pointer held in a register pair)
    typedef __attribute__((address_space(3))) uint32_t *lds_u32p;
    typedef __attribute__((address_space(3))) uint8_t *lds_u8p;
    const lds_u32p list = (lds_u32p)s_list[wave];
    const lds_u8p tags = (lds_u8p)s_tags[wave];
    int my_matches = 0;

    float F12[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    float ex = 0.f, ey = 0.f;
    if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 9; i++) F12[i] = f12all[(size_t)p * 9 + i];
        ex = epiall[2 * p]; ey = epiall[2 * p + 1];
    }
    // a9: the gates of one (query, candidate) pair that do not depend on the distance (:932-941, :970-981, :1001-1017)
    auto triang_gates = [&](int idxA, int idxB) -> bool {
        const uint32_t fA = A.mp_flags[idxA], fB = B.mp_flags[idxB];
        const bool st1 = A.uright[idxA] >= 0, st2 = B.uright[idxB] >= 0;
        if ((fA & ORBM_MP_PRESENT) || (only_stereo && !st1)) return false;                // :932, :939-941
        if ((fB & ORBM_MP_PRESENT) || (only_stereo && !st2)) return false;                // :970, :979-981
        const float x2 = B.x[idxB], y2 = B.y[idxB];
        const int oct2 = B.octave[idxB];
        if (!st1 && !st2) {
            const float dex = __fsub_rn(ex, x2), dey = __fsub_rn(ey, y2);
            if (__fadd_rn(__fmul_rn(dex, dex), __fmul_rn(dey, dey)) < __fmul_rn(100.f, B.sf[oct2])) return false;  // :1007
        }
        return check_dist_epipolar_line(A.x[idxA], A.y[idxA], x2, y2, B.sigma2[oct2], F12);                        // :1014
    };

    auto accept_match = [&](int idxA, int idxB) {                      // :312-330 / :798-815 / :1021-1041
        if (BOWM_DIAG(4)) { my_matches++; return; }
        int enc = 0;                               // SPLIT: the bin travels with the match to bow_finish_kernel (index < 65536)
        if (check_ori) {
            const int bin = rot_bin(A.angle[idxA], angB[idxB]);
            atomicAdd(&hist[bin], 1);
            if (bins_in_lds) s_bin[MODE == 0 ? idxB : idxA] = (uint8_t)bin;
            if (SPLIT) enc = bin << 16;
        }
        if (MODE == 0) match[idxB] = idxA | enc; else match[idxA] = idxB | enc;
        my_matches++;
    };

    // the reference's loops as they stand, for one node: queries in order, the 64 lanes over the candidates
    auto generic_node = [&](uint32_t offA, uint32_t nA, uint32_t offB, uint32_t nB) {
        if (MODE == 2) {
            for (uint32_t qa = 0; qa < nA; qa++) {
                const int idxA = (int)A.fv_idx[offA + qa];
                if (A.mp_flags[idxA] & ORBM_MP_PRESENT) continue;                         // :932
                uint32_t q[8];
                load_row(A.desc, idxA, q);
                uint32_t best = INITKEY;
                for (uint32_t cb = 0; cb < nB; cb += 64) {
                    const uint32_t c = cb + (uint32_t)lane;
                    if (c >= nB) continue;
                    const int idxB = (int)B.fv_idx[offB + c];
                    uint32_t cd[8];
                    load_row(B.desc, idxB, cd);
                    const uint32_t d = hamming256(q, cd);
                    if (d > ORBM_TH_LOW || !triang_gates(idxA, idxB)) continue;           // :993 (first half) and the gates
                    best = min(best, (d << 16) | (0xFFFFu - c));                          // "dist > bestDist -> skip": an equal distance replaces
                }
                best = wave_min_u32(best);
                if (best != INITKEY && lane == 0) accept_match(idxA, (int)B.fv_idx[offB + (0xFFFFu - (best & 0xFFFFu))]);   // :1021
            }
            return;
        }
        for (uint32_t qa = 0; qa < nA; qa++) {
            const int idxA = (int)A.fv_idx[offA + qa];
            const uint32_t fA = A.mp_flags[idxA];
            if (!(fA & ORBM_MP_PRESENT) || (fA & ORBM_MP_BAD)) continue;                  // :263-267 / :746-749
            uint32_t q[8];
            load_row(A.desc, idxA, q);
            uint32_t best = INITKEY, second = INITKEY;
            for (uint32_t cb = 0; cb < nB; cb += 64) {
                const uint32_t c = cb + (uint32_t)lane;
                if (c >= nB) continue;
                const int idxB = (int)B.fv_idx[offB + c];
                if (MODE == 1) {
                    const uint32_t fl = B.mp_flags[idxB];
                    if (!(fl & ORBM_MP_PRESENT) || (fl & ORBM_MP_BAD)) continue;          // :766-770
                }
                if ((claimed[idxB >> 5] >> (idxB & 31)) & 1u) continue;                   // :282 / :766
                uint32_t cd[8];
                load_row(B.desc, idxB, cd);
                const uint32_t key = (hamming256(q, cd) << 16) | c;
                second = umed3(best, second, key);
                best = min(best, key);
            }
            wave_top2(best, second);
            const int bd = (int)(best >> 16), sd = (int)(second >> 16);
            if (bd <= thr && (float)bd < __fmul_rn(ratio, (float)sd)) {                    // :304-308 / :792-796
                const int idxB = (int)B.fv_idx[offB + (best & 0xFFFFu)];
                if (lane == 0) {
                    atomicOr(&claimed[idxB >> 5], 1u << (idxB & 31));
                    accept_match(idxA, idxB);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    };

    const int first_node = part * 4 + wave, node_step = 4 * nparts;
    for (int kb = 0; first_node + node_step * kb < A.n_nodes; kb += 64) {
        // 64 of this wave's nodes, one per lane: the common-node lookup (lower_bound walk of :248-352) runs in parallel
        const int ia = first_node + node_step * (kb + lane);
        uint32_t nA = 0, nB = 0, offA = 0, offB = 0;
        if (ia < A.n_nodes) {
            // two keyframes usually populate the same nodes, so the node sits at the same index of both lists: its offsets are
            // requested together with the two node ids (one round trip to memory instead of two) and read again only on a miss
            const int guess = min(ia, B.n_nodes - 1);
            const uint32_t node_id = A.fv_node[ia], a0 = A.fv_off[ia], a1 = A.fv_off[ia + 1];
            uint32_t bn = ~node_id, b0 = 0, b1 = 0;
            if (guess >= 0) { bn = B.fv_node[guess]; b0 = B.fv_off[guess]; b1 = B.fv_off[guess + 1]; }
            int ib = guess;
            if (bn != node_id) {
                ib = find_node(B.fv_node, B.n_nodes, node_id);
                if (ib >= 0) { b0 = B.fv_off[ib]; b1 = B.fv_off[ib + 1]; }
            }
            if (ib >= 0) {
                offA = a0; nA = a1 - a0;
                offB = b0; nB = b1 - b0;
                if (nB == 0) nA = 0;
            }
        }
        unsigned long long todo = __ballot(nA != 0);
        BOWM_STAMP(2);
        // nodes (bit = lane that holds the node) none of whose records shares a query or a candidate with another record of
        // the node, found while the records are made: each of them decides for itself (resolve)
        unsigned long long simple_nodes = 0;
        bool node_contested = false;
        int node_tiles_hit = 0;
        // nodes too large for the record format: the generic loop, first
        for (unsigned long long big = __ballot(nA > REC_MAX_Q || nB > REC_MAX_T); big; big &= big - 1) {
            const int node = __builtin_ctzll(big);
            todo &= ~(1ull << node);
            generic_node((uint32_t)__builtin_amdgcn_readlane((int)offA, node), (uint32_t)__builtin_amdgcn_readlane((int)nA, node),
                         (uint32_t)__builtin_amdgcn_readlane((int)offB, node), (uint32_t)__builtin_amdgcn_readlane((int)nB, node));
        }
        // ---- the 32 x 32 tiles of the remaining nodes.  The lanes (one node each) first write the wave's TILE LIST into LDS:
        // two words per tile -- first row of the query operand | live rows << 16 | node's lane << 22 | last tile of its
        // node << 28, and first row of the candidate operand | live rows << 16 -- a node's tiles consecutive, query tiles
        // outer, candidate tiles inner.  The loop over the list then has two scalars of state per tile (where a state
        // machine over nodes, tile rows and tile columns kept some twenty, most of them in spill lanes), and the operands
        // of the NEXT tile are requested as soon as the current tile's rows have been expanded.
        uint2 *const tiles = s_tiles[wave];
        // (the list's LDS address as a scalar: entries are read at a uniform address, one v_mov away from it)
        typedef __attribute__((address_space(3))) const uint32_t *lds_u32;
        const uint32_t tiles_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint2 *)tiles);
        auto entry = [&](uint32_t i) -> uint2 {
            const lds_u32 q = (lds_u32)(uintptr_t)(tiles_lds + 8u * i);
            return make_uint2(q[0], q[1]);
        };
        uint32_t t_incl = ((todo >> lane) & 1ull) ? ((nA + 31) >> 5) * ((nB + 31) >> 5) : 0u;
        const bool listed = t_incl != 0u;
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)t_incl, dlt, 64);
            if (lane >= dlt) t_incl += o;
        }
        const uint32_t t_total = (uint32_t)__builtin_amdgcn_readlane((int)t_incl, 63);
        // lane (col, h) takes 16-byte half h of row col: dword s of it feeds k step s, for queries and candidates alike (the
        // order of the 256 bit positions inside the product is free as long as both operands use the same one)
        auto request = [&](uint32_t w0, uint32_t w1, v4u &xq, v4u &xc) {
            // a tile that starts near a node's end reads on into the next node's rows (the image ends with 32 spare rows); they
            // multiply as 0.0, see `live` below
            // (the tile's first row is wave-uniform: it moves the base on the scalar unit, and the lane's own 32 col + 16 h
            // bytes are one loop-invariant register -- no vector address arithmetic per tile)
            // (the empty asm keeps the offset's zero-extension inside this block, where the instruction selector can fold it
            // into the load: global_load_dwordx4 v, v_offset, s[base] -- hoisted out of the loop it becomes a 64-bit register
            // pair and a 64-bit vector add per operand)
            uint32_t lo32 = lane_off;                                 // 32 col + 16 h
            asm volatile("" : "+v"(lo32));
            xq = *(gptr<const v4u>)(uniform_ptr((gptr<const char>)(rowsA + 2 * (size_t)(w0 & 0xFFFFu))) + lo32);
            xc = *(gptr<const v4u>)(uniform_ptr((gptr<const char>)(rowsB + 2 * (size_t)(w1 & 0xFFFFu))) + lo32);
        };

        // ---- the claims, on the records alone.  Records wait in the wave's list (a node's records are consecutive there)
        // and are settled many nodes at a time, one record per lane: the dependent loads behind a record (keypoint
        // indices, map-point flags, angles) are then paid once per 64 records, not once per node.
        auto resolve = [&](int total) {
            if (BOWM_DIAG(1)) return;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // the records were written by other lanes of this wave
            __builtin_amdgcn_wave_barrier();
            int start = 0;
            while (start < total) {
                int take = min(64, total - start);
                uint32_t rec = 0xFFFFFFFFu;
                int tg = -1;
                if (lane < take) { rec = list[start + lane]; tg = (int)tags[start + lane]; }
                if (start + 64 < total) {                         // never split a node between two rounds (a node has <= 64 records)
                    const int next_tag = (int)tags[start + 64], last_tag = __builtin_amdgcn_readlane(tg, 63);
                    if (next_tag == last_tag) take = __builtin_ctzll(__ballot(tg == last_tag));
                }
                const bool in = lane < take;
                const int src = in ? tg : 0;
                const uint32_t nOffA = (uint32_t)__shfl((int)offA, src, 64), nOffB = (uint32_t)__shfl((int)offB, src, 64);
                const uint32_t nNA = (uint32_t)__shfl((int)nA, src, 64), nNB = (uint32_t)__shfl((int)nB, src, 64);
                const uint32_t rq = rec >> 20, rt = rec & 0x1FFFu;
                const int rd = (int)((rec >> 13) & 0x7Fu);
                int idxA = 0, idxB = 0;
                bool valid = false;
                // (rows past a node's end cannot produce records; the bounds test only keeps a logic error from becoming a wild read)
                if (in && rq < nNA && rt < nNB) {
                    idxA = (int)A.fv_idx[nOffA + rq];
                    idxB = (int)B.fv_idx[nOffB + rt];
                    if (MODE == 2) {
                        valid = triang_gates(idxA, idxB);
                    } else {
                        const uint32_t fA = A.mp_flags[idxA];
                        valid = (fA & ORBM_MP_PRESENT) && !(fA & ORBM_MP_BAD);              // :263-267 / :746-749
                        if (MODE == 1) {
                            const uint32_t fB = B.mp_flags[idxB];
                            valid = valid && (fB & ORBM_MP_PRESENT) && !(fB & ORBM_MP_BAD);  // :766-770
                        }
                    }
                }
                start += take;
                const unsigned long long vmask = __ballot(valid);
                if (!vmask) continue;
                if (MODE != 2) {
                    // the usual round: every node in it is known to have no two records with a query or a candidate in common
                    const bool simple = in && ((simple_nodes >> (tg & 63)) & 1ull) != 0ull;
                    if (!__any(valid && !simple)) {
                        if (valid && rd <= thr) accept_match(idxA, idxB);      // best = this record, second best >= capd
                        continue;
                    }
                }
                // the lanes of my node: [first, last]
                const int prev_tg = __shfl_up(tg, 1, 64);
                const unsigned long long heads = __ballot(in && (lane == 0 || tg != prev_tg));
                // (lane_r: the lane number again, opaque, so that the per-lane 64-bit masks below are made here and not kept in
                // register pairs across the tile loop)
                int lane_r = lane;
                asm volatile("" : "+v"(lane_r));
                const int first = 63 - __builtin_clzll((heads & (~0ull >> (63 - lane_r))) | 1ull);
                const unsigned long long later = lane_r < 63 ? (heads | (take < 64 ? 1ull << take : 0ull)) >> (lane_r + 1) : 0ull;
                const int last = later ? lane + __builtin_ctzll(later) : take - 1;
                const int run = last - first + 1;
                if (MODE == 2) {
                    // a9: queries are independent; per (node, query) the record of least distance, the last position among equals
                    const int lo2 = in ? first : lane, hi2 = in ? last : lane;
                    const uint32_t key2 = (rq << 20) | ((uint32_t)rd << 13) | (0x1FFFu - rt);
                    bool open2 = valid;
                    while (__any(open2)) {
                        uint32_t x = open2 ? key2 : 0xFFFFFFFFu;
#pragma unroll
                        for (int dlt = 1; dlt < 64; dlt <<= 1) {
                            const uint32_t o = (uint32_t)__shfl_up((int)x, dlt, 64);
                            if (lane - dlt >= lo2) x = min(x, o);
                        }
                        const uint32_t b = (uint32_t)__shfl((int)x, hi2, 64);
                        if (open2 && key2 == b) accept_match(idxA, idxB);
                        if (open2 && rq == (b >> 20)) open2 = false;
                    }
                    continue;
                }
                // does another valid record of my node name my query or my candidate?  (cyclic walk over the node's lanes)
                int max_run = in ? run : 0;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) max_run = max(max_run, __shfl_xor(max_run, off, 64));
                max_run = __builtin_amdgcn_readfirstlane(max_run);
                const uint32_t packed = (rq << 13) | rt | (valid ? 1u << 31 : 0u);
                bool shared = false;
                for (int dlt = 1; dlt < max_run; dlt++) {
                    int partner = lane + dlt;
                    if (partner > last) partner = first + (partner - last - 1) % max(run, 1);
                    const uint32_t o = (uint32_t)__shfl((int)packed, in ? partner : lane, 64);
                    if (in && partner != lane && (o >> 31) && (((o >> 13) & 0xFFFu) == rq || (o & 0x1FFFu) == rt)) shared = true;
                }
                const unsigned long long smask = __ballot(valid && shared);
                const unsigned long long mine_mask = run >= 64 ? ~0ull : ((1ull << run) - 1ull) << first;
                const bool contested_node = in && (smask & mine_mask) != 0ull;
                // a node where every query has one relevant candidate that nobody else wants: best = that one, second best >= capd
                if (valid && !contested_node && rd <= thr) accept_match(idxA, idxB);
                if (!smask) continue;
                // the other nodes: each takes its queries in the reference's order (ascending position), a claim takes the
                // candidate away from the node's later queries; the nodes of the round advance side by side.  A record word
                // (q << 20 | d << 13 | t) orders by query first, then by the scan's ranking (distance, position).
                const int lo = in ? first : lane, hi = in ? last : lane;
                auto node_min = [&](uint32_t x) -> uint32_t {       // minimum over the lanes of my node, in every one of them
#pragma unroll
                    for (int dlt = 1; dlt < 64; dlt <<= 1) {
                        const uint32_t o = (uint32_t)__shfl_up((int)x, dlt, 64);
                        if (lane - dlt >= lo) x = min(x, o);
                    }
                    return (uint32_t)__shfl((int)x, hi, 64);
                };
                bool open = valid && contested_node;               // record of a query not yet decided, candidate still free
                while (__any(open)) {
                    const uint32_t b = node_min(open ? rec : 0xFFFFFFFFu);                       // the node's next query and its best candidate
                    const bool mine = open && (rec >> 20) == (b >> 20);
                    const uint32_t s2 = node_min(mine && rec != b ? rec : 0xFFFFFFFFu);          // its second best ((d, t) pairs of one query are distinct)
                    if (b != 0xFFFFFFFFu) {
                        const int bd = (int)((b >> 13) & 0x7Fu);
                        const bool second_passes = s2 == 0xFFFFFFFFu || (float)bd < __fmul_rn(ratio, (float)(int)((s2 >> 13) & 0x7Fu));
                        if (bd <= thr && second_passes) {                                        // :304-308 / :792-796
                            if (open && rec == b) accept_match(idxA, idxB);
                            if (open && rt == (b & 0x1FFFu)) open = false;                       // :312 / :799: the candidate is taken
                        }
                    }
                    if (mine) open = false;
                }
            }
        };

        int node_start = 0, n_rec = 0;                        // records of finished nodes waiting in the list / records so far
        for (uint32_t t_base = 0; t_base < t_total; t_base += TL_CAP) {
        const uint32_t t_cnt = min((uint32_t)TL_CAP, t_total - t_base);
        {
            // my node's tiles that fall into this pass
            const uint32_t tBn = (nB + 31) >> 5;
            const uint32_t t_excl = listed ? t_incl - ((nA + 31) >> 5) * tBn : t_incl;
            const uint32_t k0 = max(t_excl, t_base), k1 = min(t_incl, t_base + t_cnt);
            if (listed && k0 < k1) {
                uint32_t ia = (k0 - t_excl) / tBn, ib = (k0 - t_excl) - ia * tBn;
                for (uint32_t kk = k0; kk < k1; kk++) {
                    const uint32_t qa0 = ia * 32u, cb0 = ib * 32u;
                    tiles[kk - t_base] = make_uint2((offA + qa0) | (min(32u, nA - qa0) << 16) | ((uint32_t)lane << 22) | (kk + 1 == t_incl ? 1u << 28 : 0u),
                                                    (offB + cb0) | (min(32u, nB - cb0) << 16));
                    if (++ib == tBn) { ib = 0; ia++; }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // the list is read by every lane
            __builtin_amdgcn_wave_barrier();
        }
        v4u nxq = {0, 0, 0, 0}, nxc = {0, 0, 0, 0};
        uint32_t nw0, nw1;
        {
            const uint2 e = entry(0);
            nw0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.x); nw1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.y);
            request(nw0, nw1, nxq, nxc);
        }
        for (uint32_t ti = 0; ti < t_cnt; ti++) {
            const uint32_t cw0 = nw0, cw1 = nw1;
            const int d_node = (int)((cw0 >> 22) & 63u);
            const bool d_last = (cw0 >> 28) & 1u;
            {
                v16f acc = (v16f){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                const uint32_t wq[4] = {nxq.x, nxq.y, nxq.z, nxq.w}, wc[4] = {nxc.x, nxc.y, nxc.z, nxc.w};
                // a row past the node's end multiplies as 0.0 everywhere (not as the +1.0 its zero bits would expand to): its
                // dots are 0 = distance 128, never below capd, so no record can name a row that does not exist
                const uint32_t liveq = (uint32_t)col < ((cw0 >> 16) & 63u) ? ones : 0u, livec = (uint32_t)col < ((cw1 >> 16) & 63u) ? ones : 0u;
#pragma unroll
                for (int s = 0; s < 4; s += 2) {
                    v8i fc0, fc1, fq0, fq1;
                    expand_fp4x2((uint64_t)wc[s] | ((uint64_t)wc[s + 1] << 32), livec, fc0, fc1);
                    expand_fp4x2((uint64_t)wq[s] | ((uint64_t)wq[s + 1] << 32), liveq, fq0, fq1);
                    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fc0, fq0, acc, 4, 4, 0, 0, 0, 0);   // zero scale operands select the unscaled encoding (one 8-byte instruction)
                    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fc1, fq1, acc, 4, 4, 0, 0, 0, 0);
                }
                // the next tile's rows are requested into the registers this tile's rows have just left (no second register set
                // and no copies: the loads fly under this tile's products and fold and under the other five waves of the SIMD)
                if (ti + 1 < t_cnt) {
                    const uint2 en = entry(ti + 1);
                    nw0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)en.x); nw1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)en.y);
                    request(nw0, nw1, nxq, nxc);
                }
                // folded as integers (Tf >= 0: a positive float orders like its bits, a negative one is a negative integer):
                // v_max3_i32 takes MFMA results as they are, fmaxf canonicalises each one first
                int m = max(max(__float_as_int(acc[0]), __float_as_int(acc[1])), __float_as_int(acc[2]));
#pragma unroll
                for (int r = 3; r < 15; r += 2) m = max(max(m, __float_as_int(acc[r])), __float_as_int(acc[r + 1]));
                m = max(m, __float_as_int(acc[15]));
                if (!BOWM_DIAG(2) && __any(m > Ti)) {
                    // the lane holds query qa0 + col against candidates cb0 + (r & 3) + 8 (r >> 2) + 4 h
                    const uint32_t qa0 = (cw0 & 0xFFFFu) - (uint32_t)__builtin_amdgcn_readlane((int)offA, d_node);
                    const uint32_t cb0 = (cw1 & 0xFFFFu) - (uint32_t)__builtin_amdgcn_readlane((int)offB, d_node);
                    const uint32_t qbits = (qa0 + (uint32_t)col) << 20, t0 = cb0 + (uint32_t)(4 * h);
                    const bool hitl = m > Ti;
                    const unsigned long long hm_all = __ballot(hitl);
                    const int nhit = (int)__popcll(hm_all);
                    bool hr[16];
                    int total = 0;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        hr[r] = __float_as_int(acc[r]) > Ti;
                        total += (int)__popcll(__ballot(hr[r]));
                    }
                    node_tiles_hit++;
                    if (total == nhit) {
                        // the usual tile: no lane has two hits, so a lane's hit is its maximum and its register is read off
                        // the sixteen compare masks (one select each: the scalar unit is as busy as the vector unit in this
                        // kernel, so not by OR-ing the masks there); every record of the tile in one go
                        uint32_t code = 0;                                   // (r & 3) + 8 (r >> 2) of the lane's hit
#pragma unroll
                        for (int r = 1; r < 16; r++) code = hr[r] ? (uint32_t)((r & 3) + 8 * (r >> 2)) : code;
                        const uint32_t d13 = (uint32_t)(256 - (int)__int_as_float(m)) << 12;     // distance << 13 (the dot is even)
                        const int pos = n_rec + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(hm_all >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm_all, 0u));
                        if (hitl && pos < BM_LIST) { list[pos] = (qbits | d13) + (t0 + code); tags[pos] = (uint8_t)d_node; }
                        n_rec += nhit;
                        // two records of this tile with one query (its two lanes) or one candidate (the bit was set before)?
                        const uint32_t bit = 1u << (code + (uint32_t)(4 * h));
                        uint32_t old = 0;
                        if (hitl) old = atomicOr(&s_cand[wave], bit);
                        const bool dup = __any(hitl && (old & bit) != 0u) || ((uint32_t)hm_all & (uint32_t)(hm_all >> 32)) != 0u;
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        if (lane == 0) s_cand[wave] = 0;
                        if (dup || node_tiles_hit > 1) node_contested = true;   // (records of two tiles of one node: not looked at, taken as shared)
                    } else {
                        node_contested = true;
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const unsigned long long hm = __ballot(hr[r]);
                            if (!hm) continue;
                            if (hr[r]) {
                                const uint32_t d = (uint32_t)((256 - (int)acc[r]) >> 1);
                                const int pos = n_rec + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));   // the wave counts its own records (mbcnt: no per-lane 64-bit mask to keep in registers)
                                if (pos < BM_LIST) { list[pos] = qbits | (d << 13) | (t0 + (uint32_t)((r & 3) + 8 * (r >> 2))); tags[pos] = (uint8_t)d_node; }
                            }
                            n_rec += (int)__popcll(hm);
                        }
                    }
                }
            }
            if (!d_last) continue;
            const bool was_contested = node_contested;
            node_contested = false; node_tiles_hit = 0;
            if (n_rec - node_start > 64) {
                // more relevant pairs than lanes: this node takes the generic loop, its records are dropped
                generic_node((uint32_t)__builtin_amdgcn_readlane((int)offA, d_node), (uint32_t)__builtin_amdgcn_readlane((int)nA, d_node),
                             (uint32_t)__builtin_amdgcn_readlane((int)offB, d_node), (uint32_t)__builtin_amdgcn_readlane((int)nB, d_node));
                n_rec = node_start;
            } else {
                if (!was_contested) simple_nodes |= 1ull << d_node;
                node_start = n_rec;
            }
            if (node_start > BM_LIST - 64) {                   // the next node may need 64 slots
                resolve(node_start);
                node_start = 0; n_rec = 0;
            }
        }
        __builtin_amdgcn_wave_barrier();                      // (the list is rewritten by the next pass)
        }
        BOWM_STAMP(3);
        resolve(node_start);                                  // before the lanes' node table moves on to the next 64 nodes
        BOWM_STAMP(4);
    }
    if (my_matches) atomicAdd(&nmatches, my_matches);
    __syncthreads();
    if (SPLIT) {
        if (tid < ORBM_HISTO_LENGTH && hist[tid]) atomicAdd(&pair_acc[p * 32 + tid], hist[tid]);
        if (tid == 0 && nmatches) atomicAdd(&pair_acc[p * 32 + 30], nmatches);
        return;
    }
    const int found = nmatches;                           // read by every thread before anyone changes it (the barrier inside comes first)
    if (check_ori && found > 0) {                         // a pair without matches has nothing to filter
        if (tid == 0) {
            int i1, i2, i3;
            three_maxima(hist, i1, i2, i3);
            keep[0] = i1; keep[1] = i2; keep[2] = i3;
        }
        __syncthreads();
        const int k1 = keep[0], k2 = keep[1], k3 = keep[2];
        int removed = 0;
        if (bins_in_lds) {
            // (a row is matched at most once -- vpMapPointMatches[idxF] / vpMatches12[idx1] are claimed -- so its bin is the one kept here)
            for (int i = tid; i < nOut; i += 256) {
                const int bin = (int)s_bin[i];
                if (bin != 0xFF && bin != k1 && bin != k2 && bin != k3) { match[i] = -1; removed++; }   // :368-381 / :836-849
            }
        } else
        for (int i = tid; i < nOut; i += 256) {
            const int m = match[i];
            if (m < 0) continue;
            const int idxA = MODE == 0 ? m : i, idxB = MODE == 0 ? i : m;
            const int bin = rot_bin(A.angle[idxA], angB[idxB]);
            if (bin != k1 && bin != k2 && bin != k3) { match[i] = -1; removed++; }        // :368-381 / :836-849
        }
        if (removed) atomicSub(&nmatches, removed);
        __syncthreads();
    }
    if (tid == 0) counts[p] = nmatches;
    BOWM_STAMP(5);
}

}  // namespace

// smallest distance that passes the ratio test as second best for every acceptable best distance (see the header comment);
// never below TH_LOW + 1, so that every acceptable best distance is "relevant" too
int orbm_bow_cap(float ratio) {
    if (!(ratio > 0.f)) return 257;
    int c = ORBM_TH_LOW + 1;
    while (c <= 256 && !((float)c * ratio > (float)ORBM_TH_LOW)) c++;
    return c;
}

#ifdef ORBM_DIAG
extern "C" void orbm__bowm_diag(int bits) { (void)hipMemcpyToSymbol(HIP_SYMBOL(d_bowm_diag), &bits, sizeof(int)); }
extern "C" void orbm__bowm_trace(unsigned long long *out) { (void)hipDeviceSynchronize(); (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(d_bowm_trace), sizeof(unsigned long long) * 64 * 8); }
#endif

int orbm_launch_bowm(orbm_ctx *ctx, int mode, int n_pairs, const int32_t *dfa, const int32_t *dfb, float ratio, int capd,
                     int check_ori, int stride, int32_t *dm, int32_t *dc, int splits, int *pair_acc, int only_stereo,
                     const float *df12, const float *dep) {
#define LAUNCH(M, S)                                                                                                      \
    hipLaunchKernelGGL((bowm_kernel<M, S>), dim3((unsigned)(n_pairs * (S ? splits : 1))), dim3(256), 0, ctx->stream, ctx->d_frames, dfa, \
                       dfb, ratio, capd, check_ori, stride, dm, dc, splits, pair_acc, only_stereo, df12, dep)
    if (splits > 1) { if (mode == 0) LAUNCH(0, true); else if (mode == 1) LAUNCH(1, true); else LAUNCH(2, true); }
    else { if (mode == 0) LAUNCH(0, false); else if (mode == 1) LAUNCH(1, false); else LAUNCH(2, false); }
#undef LAUNCH
    HIPCHK(ctx, hipGetLastError());
    return ORBM_OK;
}
