// window.hip -- grid-cell windowed matching with in-order claims:
//   a6  SearchByProjection(Frame&, vector<MapPoint*>&, th)             ORBmatcher.cc:57-166     (MODE_PROJ)
//   a11 SearchByProjection(Cur, Last, th, mono)                        ORBmatcher.cc:1768-1950  (MODE_LAST)
//       SearchByProjection(Cur, KF, sAlreadyFound, th, ORBdist)        ORBmatcher.cc:1963-2123  (MODE_KF)
//       SearchByProjection(KF, Scw, vpPoints, vpMatched, th)           ORBmatcher.cc:401-542    (MODE_SIM3)
//   a5  SearchForInitialization                                       ORBmatcher.cc:562-689    (MODE_INIT)
// on top of Frame::GetFeaturesInArea / KeyFrame::GetFeaturesInArea (a13, a14: Frame.cc:434-498,
// KeyFrame.cc:766-812) walked on the device-resident grid CSR.
//
// The reference processes queries one after the other and a query's accepted match changes what later
// queries may see (F.mvpMapPoints[idx], vpMatched[idx], vMatchedDistance[idx]).  To run queries in
// parallel and still return exactly the sequential result, one workgroup owns one (frame, query set)
// instance, keeps the per-keypoint slot state in LDS and takes the queries in index order, one chunk of
// blockDim queries at a time (one query per lane, its descriptor and window in registers, read from HBM
// once).  Inside a chunk it iterates rounds:
//   evaluate : every unresolved query walks its window in reference order against the current slot state
//              and posts its chunk-local index with atomicMin on every slot it could still claim
//              (distance <= acceptance threshold);
//   decide   : a query is FINAL iff no lower-index unresolved query posted on any slot of its window;
//              final queries apply their result to the slot state.
// A query can only ever claim a slot it posted on, so a final query has seen the exact state the
// sequential loop would have shown it; the lowest unresolved query is always final, so every round
// makes progress.  Typical inputs need 2-4 rounds per chunk.
#include <limits.h>
#include <string.h>
#include <algorithm>
#include "orbm_internal.h"

namespace {

enum { MODE_LAST = 0, MODE_KF = 1, MODE_SIM3 = 2, MODE_INIT = 3, MODE_PROJ = 10 };

constexpr int WIN_MAX_N = 8192;       // slot state lives in LDS: 5 x 4 B per keypoint
constexpr uint32_t NOPOST = 0xFFFFFFFFu;

struct WinParams {
    const FrameDev *frames;
    const QSetDev *qsets;
    const int32_t *frame_ids;
    const int32_t *qset_ids;     // MODE_INIT: id of frame 1 (the query frame)
    const uint8_t *slot_flags;   // nullable, n_inst x stride
    float *prev_xy;              // MODE_INIT only
    int32_t *out;                // owner (n_inst x stride) | MODE_INIT: matches12
    int32_t *counts;
    float th, ratio;
    int dist_th, check_ori, stride, window;
};

// Read-only frame data the walk and the distance loop touch, either in global memory or staged in LDS.
struct FrameAcc {
    const uint32_t *cell_off;   // 3073
    const float4 *rec;          // cell-ordered {x, y, uright, idx | oct << 16}
    const uint4 *desc;          // 2 per keypoint
    float min_x, min_y, inv_w, inv_h;
};

// Frame::GetFeaturesInArea (Frame.cc:434-498) / KeyFrame::GetFeaturesInArea (KeyFrame.cc:766-812):
// calls body(k, octave, uright) for every keypoint the reference would push, in the reference's order
// (ix ascending, iy ascending, insertion order).  Cells (ix, minY..maxY) are consecutive in the CSR, so a
// column of the window is ONE contiguous run of 16-byte records.
template <bool KF_MODE, typename Body>
__device__ __forceinline__ void walk_window(const FrameAcc &f, float x, float y, float r, int minLevel, int maxLevel,
                                            Body &&body) {
    const float xr = __fsub_rn(x, f.min_x), yr = __fsub_rn(y, f.min_y);
    const int nMinCellX = max(0, (int)floorf(__fmul_rn(__fsub_rn(xr, r), f.inv_w)));
    if (nMinCellX >= ORBM_GRID_COLS) return;
    const int nMaxCellX = min(ORBM_GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(xr, r), f.inv_w)));
    if (nMaxCellX < 0) return;
    const int nMinCellY = max(0, (int)floorf(__fmul_rn(__fsub_rn(yr, r), f.inv_h)));
    if (nMinCellY >= ORBM_GRID_ROWS) return;
    const int nMaxCellY = min(ORBM_GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(yr, r), f.inv_h)));
    if (nMaxCellY < 0) return;
    const bool bCheckLevels = !KF_MODE && ((minLevel > 0) || (maxLevel >= 0));
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
        const uint32_t b = f.cell_off[ix * ORBM_GRID_ROWS + nMinCellY];
        const uint32_t e = f.cell_off[ix * ORBM_GRID_ROWS + nMaxCellY + 1];
        for (uint32_t j = b; j < e; j++) {
            const float4 rc = f.rec[j];
            const uint32_t w = __float_as_uint(rc.w);
            const int oct = (int)(w >> 16);
            if (bCheckLevels) {
                if (oct < minLevel) continue;
                if (maxLevel >= 0 && oct > maxLevel) continue;
            }
            const float distx = __fsub_rn(rc.x, x), disty = __fsub_rn(rc.y, y);
            if (fabsf(distx) < r && fabsf(disty) < r) body((int)(w & 0xFFFFu), oct, rc.z);
        }
    }
}

constexpr int WIN_CACHE = 16;  // candidates remembered per query (registers); wider windows fall back to re-walking

template <int MODE, bool LDS_FRAME>
__global__ void window_kernel(WinParams P) {
    extern __shared__ __align__(16) uint32_t lds[];
    const int inst = blockIdx.x;
    const FrameDev &F = P.frames[P.frame_ids[inst]];
    const int n = F.n;
    const int tid = threadIdx.x, nthr = blockDim.x;
    // LDS carve.  LDS_FRAME: [cell_off 3076][rec 4n][desc 8n] then the slot state; else slot state only.
    FrameAcc acc;
    acc.min_x = F.min_x; acc.min_y = F.min_y; acc.inv_w = F.inv_w; acc.inv_h = F.inv_h;
    uint32_t *base = lds;
    if (LDS_FRAME) {
        uint32_t *l_cell = lds;
        float4 *l_rec = reinterpret_cast<float4 *>(lds + 3076);
        uint4 *l_desc = reinterpret_cast<uint4 *>(lds + 3076 + 4 * (size_t)n);
        for (int i = tid; i <= ORBM_GRID_CELLS; i += nthr) l_cell[i] = F.cell_off[i];
        for (int i = tid; i < n; i += nthr) l_rec[i] = F.cell_rec[i];
        const uint4 *gd = reinterpret_cast<const uint4 *>(F.desc);
        for (int i = tid; i < 2 * n; i += nthr) l_desc[i] = gd[i];
        acc.cell_off = l_cell; acc.rec = l_rec; acc.desc = l_desc;
        base = lds + 3076 + 12 * (size_t)n;
    } else {
        acc.cell_off = F.cell_off; acc.rec = F.cell_rec; acc.desc = reinterpret_cast<const uint4 *>(F.desc);
    }
    uint32_t *state = base;                       // ORBM_MP_* flags | vMatchedDistance
    int32_t *owner = (int32_t *)(base + n);
    uint32_t *post0 = base + 2 * (size_t)n, *post1 = base + 3 * (size_t)n;
    uint32_t *binmask = base + 4 * (size_t)n;
    int *hist = (int *)(base + 5 * (size_t)n);
    int *nmatch = hist + 32;
    int *keep = hist + 33;
    int *bucket = hist + 64;                                    // 32 cost classes
    uint16_t *perm = reinterpret_cast<uint16_t *>(hist + 96);   // blockDim entries

    // query source
    const QSetDev *QS = nullptr;
    const FrameDev *F1 = nullptr;
    int Q;
    if (MODE == MODE_INIT) { F1 = &P.frames[P.qset_ids[inst]]; Q = F1->n; }
    else { QS = &P.qsets[P.qset_ids[inst]]; Q = QS->n; }
    int32_t *out = P.out + (size_t)inst * P.stride;

    for (int i = tid; i < n; i += nthr) {
        uint32_t s0;
        if (MODE == MODE_INIT) s0 = (uint32_t)INT_MAX;                                   // vMatchedDistance (:573)
        else s0 = P.slot_flags ? P.slot_flags[(size_t)inst * P.stride + i] : F.mp_flags[i];
        state[i] = s0;
        owner[i] = -1;
        post0[i] = NOPOST; post1[i] = NOPOST;
        binmask[i] = 0;
    }
    if (MODE == MODE_INIT) for (int i = tid; i < Q; i += nthr) out[i] = -1;              // vnMatches12 (:566)
    if (tid < 32) hist[tid] = 0;
    if (tid == 0) *nmatch = 0;
    __syncthreads();

    const bool use_hist = P.check_ori && (MODE == MODE_LAST || MODE == MODE_KF || MODE == MODE_INIT) &&
                          (MODE == MODE_INIT || QS->angle != nullptr);
    const bool bFactor = P.th != 1.0f;                                                     // :61
    const bool stereo_gate = MODE == MODE_PROJ || (MODE == MODE_LAST && QS->f3 != nullptr);
    const int thr = MODE == MODE_PROJ ? ORBM_TH_HIGH : (MODE == MODE_INIT ? ORBM_TH_LOW : P.dist_th);
    int my_matches = 0;

    // One query per lane.  Window areas differ by more than 10x between pyramid levels and a wave costs what its
    // widest window costs, so the lane -> query assignment inside a chunk is sorted by window size (counting sort
    // over 32 classes of covered grid cells).  Claim priority stays the query's index (`ql`, chunk-local), so the
    // result is the sequential one whatever the assignment.  The NEXT chunk is sorted and loaded while the current
    // one is matched.
    struct Query {
        uint4 d0, d1;
        float x, y, r, xr, angle;
        int minL, maxL, ql;
        uint32_t flags;
        bool valid;
    };
    auto load_params = [&](int qi) {
        Query q;
        q.d0 = q.d1 = make_uint4(0, 0, 0, 0);
        q.x = q.y = q.r = q.xr = q.angle = 0.f; q.minL = q.maxL = -1; q.ql = 0; q.flags = 0; q.valid = false;
        if (qi >= Q) return q;
        if (MODE == MODE_INIT) {
            if (F1->octave[qi] == 0) {                                                    // :582-584
                q.valid = true;
                q.x = P.prev_xy[2 * qi]; q.y = P.prev_xy[2 * qi + 1];
                q.r = (float)P.window; q.minL = 0; q.maxL = 0;                            // :587
                q.angle = F1->angle[qi];
            }
        } else {
            q.flags = QS->flags[qi];
            if (q.flags & ORBM_Q_VALID) {
                q.valid = true;
                q.x = QS->f0[qi]; q.y = QS->f1[qi];
                if (MODE == MODE_PROJ) {
                    const int lvl = QS->l0[qi];
                    float rv = ((double)QS->f3[qi] > 0.998) ? 2.5f : 4.0f;              // :168-174
                    if (bFactor) rv = __fmul_rn(rv, P.th);                               // :84-85
                    q.r = __fmul_rn(rv, F.sf[lvl]);                                      // :89
                    q.minL = lvl - 1; q.maxL = lvl;
                    q.xr = QS->f2[qi];
                } else {
                    q.r = QS->f2[qi]; q.minL = QS->l0[qi]; q.maxL = QS->l1[qi];
                    if (QS->f3) q.xr = QS->f3[qi];
                    if (QS->angle) q.angle = QS->angle[qi];
                }
            }
        }
        return q;
    };
    auto load_query = [&](int qbase, int ql) {
        Query q = load_params(qbase + ql);
        q.ql = ql;
        if (q.valid) {
            const uint32_t *dsrc = MODE == MODE_INIT ? F1->desc : QS->desc;
            const uint4 *p4 = reinterpret_cast<const uint4 *>(dsrc) + 2 * (size_t)(qbase + ql);
            q.d0 = p4[0]; q.d1 = p4[1];
        }
        return q;
    };
    // cost class of a query = grid cells its window covers / 4 (0 = nothing to do)
    auto cost_class = [&](const Query &q) {
        if (!q.valid) return 0;
        const float xr_ = __fsub_rn(q.x, acc.min_x), yr_ = __fsub_rn(q.y, acc.min_y);
        const int x0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(xr_, q.r), acc.inv_w)));
        const int x1 = min(ORBM_GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(xr_, q.r), acc.inv_w)));
        const int y0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(yr_, q.r), acc.inv_h)));
        const int y1 = min(ORBM_GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(yr_, q.r), acc.inv_h)));
        if (x1 < x0 || y1 < y0) return 0;
        return min(31, 1 + ((x1 - x0 + 1) * (y1 - y0 + 1) >> 2));
    };
    // returns the chunk-local index of the query this lane takes (widest windows first)
    auto assign_lane = [&](int cls) {
        if (MODE == MODE_INIT) return tid;           // one radius for all queries (:587): nothing to balance
        if (tid < 32) bucket[tid] = 0;
        __syncthreads();
        const int rank = atomicAdd(&bucket[cls], 1);
        __syncthreads();
        int start = 0;
        for (int b = 31; b > cls; b--) start += bucket[b];
        perm[start + rank] = (uint16_t)tid;
        __syncthreads();
        return (int)perm[tid];
    };

    Query pnext = load_params(tid);
    Query nextq = load_query(0, assign_lane(cost_class(pnext)));
    pnext = load_params(nthr + tid);
    for (int base_q = 0; base_q < Q; base_q += nthr) {
        const Query cur = nextq;
        const int ql = cur.ql, qi = base_q + ql;
        if (base_q + nthr < Q) {
            nextq = load_query(base_q + nthr, assign_lane(cost_class(pnext)));
            pnext = load_params(base_q + 2 * nthr + tid);
        }
        bool unresolved = cur.valid;
        const uint32_t qd[8] = {cur.d0.x, cur.d0.y, cur.d0.z, cur.d0.w, cur.d1.x, cur.d1.y, cur.d1.z, cur.d1.w};
        const float x = cur.x, y = cur.y, r = cur.r, xr = cur.xr, qangle = cur.angle;
        const int minL = cur.minL, maxL = cur.maxL;
        const uint32_t qflags = cur.flags;

        // ---- collect: ONE window walk per query.  The static gates (level range, |dx|<r, |dy|<r, octave filter,
        // stereo gate) and the Hamming distance do not depend on the slot state, so the candidates that pass are
        // remembered as k | oct << 16 | dist << 20 and the rounds below never walk again.  More than WIN_CACHE
        // candidates (wide windows): the rounds fall back to walking.
        int ncand = 0;
        uint32_t cc[WIN_CACHE];
#pragma unroll
        for (int i = 0; i < WIN_CACHE; i++) cc[i] = 0;
        if (unresolved) {
            auto collect = [&](int k, int oct, float ur) {
                if (MODE == MODE_SIM3) { if (oct < minL || oct > maxL) return; }          // :515-518
                if (stereo_gate) {
                    if (ur > 0) {
                        const float er = fabsf(__fsub_rn(xr, ur));
                        if (er > r) return;                                                // :117-122 / :1883-1891
                    }
                }
                if (ncand < WIN_CACHE) {
                    const uint4 lo = acc.desc[2 * k], hi = acc.desc[2 * k + 1];
                    const uint32_t cd[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    const uint32_t v = (uint32_t)k | ((uint32_t)oct << 16) | (hamming256(qd, cd) << 20);
#pragma unroll
                    for (int i = 0; i < WIN_CACHE; i++) if (i == ncand) cc[i] = v;
                }
                ncand++;
            };
            if (MODE == MODE_SIM3) walk_window<true>(acc, x, y, r, minL, maxL, collect);
            else walk_window<false>(acc, x, y, r, minL, maxL, collect);
            if (ncand == 0) unresolved = false;   // empty window: nothing to match, nothing to claim (:91-92)
        }
        const bool cached = ncand <= WIN_CACHE;
        int wave_ncand = min(ncand, WIN_CACHE);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wave_ncand = max(wave_ncand, __shfl_xor(wave_ncand, off, 64));
        wave_ncand = __builtin_amdgcn_readfirstlane(wave_ncand);

        int round = 0;
        while (true) {
            uint32_t *post = (round & 1) ? post1 : post0;
            uint32_t *post_next = (round & 1) ? post0 : post1;
            int bestDist = MODE == MODE_INIT ? INT_MAX : 256, bestDist2 = bestDist, bestIdx = -1;
            int bestLevel = -1, bestLevel2 = -1;
            // the reference's inner loop body for one candidate that passed the static gates
            auto consider = [&](int k, int oct, int dist) {
                const uint32_t st = state[k];
                if (MODE == MODE_PROJ || MODE == MODE_LAST) {
                    if ((st & ORBM_MP_PRESENT) && (st & ORBM_MP_OBS)) return;              // :113-115 / :1879-1881
                } else if (MODE != MODE_INIT) {
                    if (st & ORBM_MP_PRESENT) return;                                      // :2048 / :512
                }
                if (MODE == MODE_INIT) { if ((int)st <= dist) return; }                    // :608
                if (dist <= thr) atomicMin(&post[k], (uint32_t)ql);                       // this query may still claim k
                if (dist < bestDist) {
                    bestDist2 = bestDist; bestDist = dist;
                    bestLevel2 = bestLevel; bestLevel = oct;
                    bestIdx = k;
                } else if (dist < bestDist2) {
                    bestLevel2 = oct; bestDist2 = dist;
                }
            };
            // ---- evaluate ---------------------------------------------------------------------------
            if (unresolved) {
                if (cached) {
                    // wave-uniform trip count: the register array is indexed by a scalar (v_movrels), no unrolling
                    for (int i = 0; i < wave_ncand; i++) {
                        const uint32_t v = cc[i];
                        if (i < ncand) consider((int)(v & 0xFFFFu), (int)((v >> 16) & 15u), (int)(v >> 20));
                    }
                } else {
                    auto body = [&](int k, int oct, float ur) {
                        if (MODE == MODE_SIM3) { if (oct < minL || oct > maxL) return; }
                        if (stereo_gate) {
                            if (ur > 0) {
                                const float er = fabsf(__fsub_rn(xr, ur));
                                if (er > r) return;
                            }
                        }
                        const uint4 lo = acc.desc[2 * k], hi = acc.desc[2 * k + 1];
                        const uint32_t cd[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                        consider(k, oct, (int)hamming256(qd, cd));
                    };
                    if (MODE == MODE_SIM3) walk_window<true>(acc, x, y, r, minL, maxL, body);
                    else walk_window<false>(acc, x, y, r, minL, maxL, body);
                }
            }
            __syncthreads();
            // ---- decide -----------------------------------------------------------------------------
            if (unresolved) {
                bool is_final = true;
                if (cached) {
                    for (int i = 0; i < wave_ncand; i++) {
                        const uint32_t v = cc[i];
                        if (i < ncand && post[v & 0xFFFFu] < (uint32_t)ql) is_final = false;
                    }
                } else {
                    auto chk = [&](int k, int oct, float) {
                        if (MODE == MODE_SIM3) { if (oct < minL || oct > maxL) return; }
                        if (post[k] < (uint32_t)ql) is_final = false;
                    };
                    if (MODE == MODE_SIM3) walk_window<true>(acc, x, y, r, minL, maxL, chk);
                    else walk_window<false>(acc, x, y, r, minL, maxL, chk);
                }
                if (is_final) {
                    unresolved = false;
                    bool accept;
                    if (MODE == MODE_PROJ)
                        accept = bestDist <= ORBM_TH_HIGH &&
                                 !(bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(P.ratio, (float)bestDist2));  // :153-158
                    else if (MODE == MODE_INIT)
                        accept = bestDist <= ORBM_TH_LOW && (float)bestDist < __fmul_rn((float)bestDist2, P.ratio);    // :625-627
                    else
                        accept = bestDist <= thr;                                         // :1904 / :2062 / :533
                    if (accept) {
                        if (MODE == MODE_INIT) {
                            const int old = owner[bestIdx];
                            if (old >= 0) { out[old] = -1; my_matches--; }                 // :629-633
                            out[qi] = bestIdx; owner[bestIdx] = qi; state[bestIdx] = (uint32_t)bestDist;  // :636-638
                        } else {
                            uint32_t ns = ORBM_MP_PRESENT;
                            if ((MODE != MODE_PROJ && MODE != MODE_LAST) || (qflags & ORBM_Q_OBS)) ns |= ORBM_MP_OBS;
                            state[bestIdx] = ns;
                            owner[bestIdx] = qi;                                           // :160 / :1906 / :2066 / :535
                        }
                        my_matches++;
                        if (use_hist) {
                            const int bin = rot_bin(qangle, F.angle[bestIdx]);             // :644-651 / :1911-1918 / :2079-2090
                            atomicAdd(&hist[bin], 1);
                            if (MODE != MODE_INIT) atomicOr(&binmask[bestIdx], 1u << bin);
                        }
                    }
                }
            }
            // the other post buffer is idle in this phase (decide only reads `post`): clear it for the next round
            for (int i = tid; i < n; i += nthr) post_next[i] = NOPOST;
            if (!__syncthreads_or(unresolved ? 1 : 0)) break;
            round++;
        }
        // leave both post buffers clean for the next chunk
        {
            uint32_t *post = (round & 1) ? post1 : post0;
            for (int i = tid; i < n; i += nthr) post[i] = NOPOST;
        }
        __syncthreads();
    }

    if (my_matches) atomicAdd(nmatch, my_matches);
    __syncthreads();
    if (use_hist) {
        if (tid == 0) {
            int i1, i2, i3;
            three_maxima(hist, i1, i2, i3);
            keep[0] = i1; keep[1] = i2; keep[2] = i3;
            // every accept event in a dropped bin is one nmatches-- (:1934-1946 / :2109-2119); MODE_INIT counts below
            if (MODE != MODE_INIT) {
                int drop = 0;
                for (int b = 0; b < ORBM_HISTO_LENGTH; b++)
                    if (b != i1 && b != i2 && b != i3) drop += hist[b];
                *nmatch -= drop;
            }
        }
        __syncthreads();
        const int k1 = keep[0], k2 = keep[1], k3 = keep[2];
        uint32_t keepmask = 0;
        if (k1 >= 0) keepmask |= 1u << k1;
        if (k2 >= 0) keepmask |= 1u << k2;
        if (k3 >= 0) keepmask |= 1u << k3;
        if (MODE == MODE_INIT) {
            int removed = 0;
            for (int i = tid; i < Q; i += nthr) {
                const int m = out[i];
                if (m < 0) continue;                                                      // :673 (a stolen match is already -1)
                const int bin = rot_bin(F1->angle[i], F.angle[m]);
                if (!((keepmask >> bin) & 1u)) { out[i] = -1; removed++; }
            }
            if (removed) atomicSub(nmatch, removed);
        } else {
            for (int i = tid; i < n; i += nthr)
                if (binmask[i] & ~keepmask) owner[i] = -2;                                 // slot set to NULL (:1942 / :2115)
        }
        __syncthreads();
    }
    if (MODE == MODE_INIT) {
        for (int i = tid; i < Q; i += nthr) {                                             // :684-686
            const int m = out[i];
            if (m >= 0) { P.prev_xy[2 * i] = F.x[m]; P.prev_xy[2 * i + 1] = F.y[m]; }
        }
    } else {
        for (int i = tid; i < n; i += nthr) out[i] = owner[i];
    }
    if (tid == 0) P.counts[inst] = *nmatch;
}

// a12: claim-free best-1 (SearchBySim3 / Fuse inner loops).  Queries are independent: one lane per query, grid =
// instances x query chunks, frame read through L2 (cell-ordered records stream contiguously).
__global__ __launch_bounds__(256) void free_kernel(WinParams P, int chunks_per_inst, int chi2_gate) {
    const int inst = blockIdx.x / chunks_per_inst;
    const int chunk = blockIdx.x - inst * chunks_per_inst;
    const FrameDev &F = P.frames[P.frame_ids[inst]];
    const QSetDev &QS = P.qsets[P.qset_ids[inst]];
    const int qi = chunk * 256 + (int)threadIdx.x;
    if (qi >= QS.n || qi >= P.stride) return;
    int32_t *out = P.out + (size_t)inst * P.stride;
    int result = -1;
    if (QS.flags[qi] & ORBM_Q_VALID) {
        FrameAcc acc;
        acc.min_x = F.min_x; acc.min_y = F.min_y; acc.inv_w = F.inv_w; acc.inv_h = F.inv_h;
        acc.cell_off = F.cell_off; acc.rec = F.cell_rec; acc.desc = reinterpret_cast<const uint4 *>(F.desc);
        const float u = QS.f0[qi], v = QS.f1[qi], radius = QS.f2[qi];
        const float ur = QS.f3 ? QS.f3[qi] : 0.f;
        const int minL = QS.l0[qi], maxL = QS.l1[qi];
        const uint4 *p4 = reinterpret_cast<const uint4 *>(QS.desc) + 2 * (size_t)qi;
        const uint4 d0 = p4[0], d1 = p4[1];
        const uint32_t qd[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
        int bestDist = INT_MAX, bestIdx = -1;
        auto body = [&](int k, int oct, float kur) {
            if (oct < minL || oct > maxL) return;                                          // :1207 / :1406 / :1608
            if (chi2_gate) {
                const float inv = __fdiv_rn(1.0f, F.sigma2[oct]);
                const float ex = __fsub_rn(u, F.x[k]), ey = __fsub_rn(v, F.y[k]);
                if (kur >= 0) {                                                            // :1212-1225
                    const float er = __fsub_rn(ur, kur);
                    const float e2 = __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(er, er));
                    if ((double)__fmul_rn(e2, inv) > 7.8) return;
                } else {                                                                   // :1226-1238
                    const float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
                    if ((double)__fmul_rn(e2, inv) > 5.99) return;
                }
            }
            const uint4 lo = acc.desc[2 * k], hi = acc.desc[2 * k + 1];
            const uint32_t cd[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            const int dist = (int)hamming256(qd, cd);
            if (dist < bestDist) { bestDist = dist; bestIdx = k; }
        };
        walk_window<true>(acc, u, v, radius, -1, -1, body);
        if (bestDist <= P.dist_th) result = bestIdx;
    }
    out[qi] = result;
}

constexpr int WIN_LDS_FRAME_MAX_N = 2160;   // cell_off + 16 B records + 32 B descriptors + 20 B state per keypoint + lane assignment <= 160 KB

template <int MODE, bool LDS_FRAME>
int launch_mode2(orbm_ctx *ctx, const WinParams &P, int n_inst, int max_n, int threads) {
    const size_t lds = ((size_t)(LDS_FRAME ? 3076 + 12 * (size_t)max_n : 0) + 5 * (size_t)max_n + 96 + 512) * 4;
    if (lds > 48 * 1024) {
        HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&window_kernel<MODE, LDS_FRAME>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipLaunchKernelGGL((window_kernel<MODE, LDS_FRAME>), dim3(n_inst), dim3(threads), lds, ctx->stream, P);
    HIPCHK(ctx, hipGetLastError());
    return ORBM_OK;
}

int g_win_force_global = 0;   // tuning / test knob: 1 = never stage the frame in LDS

template <int MODE>
int launch_mode(orbm_ctx *ctx, const WinParams &P, int n_inst, int max_n, int threads) {
    // the whole frame (grid, records, descriptors) fits one CU's LDS up to 2200 keypoints: 1024-thread workgroup
    if (max_n <= WIN_LDS_FRAME_MAX_N && !g_win_force_global) return launch_mode2<MODE, true>(ctx, P, n_inst, max_n, MODE == MODE_INIT ? 256 : 1024);
    return launch_mode2<MODE, false>(ctx, P, n_inst, max_n, threads);
}

int launch_window(orbm_ctx *ctx, int mode, const WinParams &P, int n_inst, int max_n) {
    const int threads = 512;
    switch (mode) {
        case MODE_LAST: return launch_mode<MODE_LAST>(ctx, P, n_inst, max_n, threads);
        case MODE_KF: return launch_mode<MODE_KF>(ctx, P, n_inst, max_n, threads);
        case MODE_SIM3: return launch_mode<MODE_SIM3>(ctx, P, n_inst, max_n, threads);
        case MODE_INIT: return launch_mode<MODE_INIT>(ctx, P, n_inst, max_n, 256);
        default: return launch_mode<MODE_PROJ>(ctx, P, n_inst, max_n, threads);
    }
}

int run_window_batch(orbm_ctx *ctx, int mode, int n_inst, const int32_t *frame, const int32_t *qset, float th,
                     float ratio, int dist_th, int check_ori, const uint8_t *slot_flags, int32_t stride,
                     int32_t *owner, int32_t *counts, int io_on_device, const char *who) {
    if (!ctx) return ORBM_EINVAL;
    if (n_inst < 0 || (n_inst > 0 && (!frame || !qset || !owner || !counts)) || stride < 0)
        return orbm_fail(ctx, ORBM_EINVAL, "%s: null argument or negative size", who);
    if (n_inst == 0) return ORBM_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int max_n = 0;
    if (!io_on_device) {
        int rc = orbm_check_frames(ctx, n_inst, frame, "frame");
        if (!rc) rc = orbm_check_qsets(ctx, n_inst, qset, mode == MODE_PROJ ? 0 : 1, "qset");
        if (rc) return rc;
        for (int k = 0; k < n_inst; k++) max_n = std::max(max_n, ctx->h_frames[frame[k]].n);
        if (max_n > stride) return orbm_fail(ctx, ORBM_EINVAL, "%s: stride %d < largest frame %d", who, stride, max_n);
    } else {
        max_n = stride;  // the caller's stride bounds every frame of the batch
    }
    if (max_n > WIN_MAX_N)
        return orbm_fail(ctx, ORBM_ELIMIT, "%s: frames of more than %d keypoints are not supported by the windowed kernels (got %d)", who, WIN_MAX_N, max_n);
    const size_t out_elems = (size_t)n_inst * stride;
    IoStage io(ctx, io_on_device);
    if (!io_on_device) {
        int rc = orbm_ws_reserve(ctx, (size_t)n_inst * 12 + out_elems * 5 + 8192);
        if (rc) return rc;
    }
    const void *dfr, *dqs, *dsf = nullptr; void *down, *dc;
    int rc;
    if ((rc = io.in(frame, (size_t)n_inst * 4, &dfr)) || (rc = io.in(qset, (size_t)n_inst * 4, &dqs)) ||
        (rc = io.in(slot_flags, out_elems, &dsf)) || (rc = io.out(owner, out_elems * 4, &down)) ||
        (rc = io.out(counts, (size_t)n_inst * 4, &dc)))
        return rc;
    if (!io_on_device && out_elems) HIPCHK(ctx, hipMemsetAsync(down, 0xFF, out_elems * 4, ctx->stream));
    WinParams P;
    memset(&P, 0, sizeof P);
    P.frames = ctx->d_frames; P.qsets = ctx->d_qsets;
    P.frame_ids = (const int32_t *)dfr; P.qset_ids = (const int32_t *)dqs;
    P.slot_flags = (const uint8_t *)dsf;
    P.out = (int32_t *)down; P.counts = (int32_t *)dc;
    P.th = th; P.ratio = ratio; P.dist_th = dist_th; P.check_ori = check_ori; P.stride = stride;
    rc = launch_window(ctx, mode, P, n_inst, std::max(max_n, 1));
    if (rc) return rc;
    return io.finish();
}

}  // namespace

// tuning / test knob (not part of the public ABI): 1 = keep the frame in global memory even when it fits LDS
extern "C" void orbm__window_force_global(int v) { g_win_force_global = v; }

extern "C" int orbm_search_proj_points_batch(orbm_ctx *ctx, int n_inst, const int32_t *frame, const int32_t *qset,
                                             float th, float ratio, const uint8_t *slot_flags, int32_t stride,
                                             int32_t *owner, int32_t *counts, int io_on_device) {
    return run_window_batch(ctx, MODE_PROJ, n_inst, frame, qset, th, ratio, ORBM_TH_HIGH, 0, slot_flags, stride, owner,
                            counts, io_on_device, "search_proj_points_batch");
}

extern "C" int orbm_search_window_batch(orbm_ctx *ctx, int mode, int n_inst, const int32_t *frame,
                                        const int32_t *qset, int dist_th, int check_ori, const uint8_t *slot_flags,
                                        int32_t stride, int32_t *owner, int32_t *counts, int io_on_device) {
    if (ctx && (mode < 0 || mode > 2)) return orbm_fail(ctx, ORBM_EINVAL, "search_window_batch: mode must be 0, 1 or 2");
    return run_window_batch(ctx, mode, n_inst, frame, qset, 1.0f, 1.0f, dist_th, check_ori, slot_flags, stride, owner,
                            counts, io_on_device, "search_window_batch");
}

extern "C" int orbm_search_init(orbm_ctx *ctx, int32_t frame1, int32_t frame2, float *prev_xy, int window, float ratio,
                                int check_ori, int32_t *matches12) {
    if (!ctx) return ORBM_EINVAL;
    int rc = orbm_check_frames(ctx, 1, &frame1, "frame1");
    if (!rc) rc = orbm_check_frames(ctx, 1, &frame2, "frame2");
    if (rc) return rc;
    const int n1 = ctx->h_frames[frame1].n, n2 = ctx->h_frames[frame2].n;
    if (n1 > 0 && (!prev_xy || !matches12)) return orbm_fail(ctx, ORBM_EINVAL, "search_init: null argument");
    if (n1 == 0) return 0;
    if (n2 > WIN_MAX_N) return orbm_fail(ctx, ORBM_ELIMIT, "search_init: frame 2 has %d keypoints (limit %d)", n2, WIN_MAX_N);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    rc = orbm_ws_reserve(ctx, (size_t)n1 * 12 + 4096);
    if (rc) return rc;
    char *w = (char *)ctx->ws;
    int32_t *d_ids = (int32_t *)w;                 // [frame2, frame1]
    int32_t *d_count = (int32_t *)(w + 256);
    float *d_prev = (float *)(w + 512);
    int32_t *d_m12 = (int32_t *)(w + 512 + (((size_t)n1 * 8 + 255) / 256) * 256);
    const int32_t ids[2] = {frame2, frame1};
    HIPCHK(ctx, hipMemcpyAsync(d_ids, ids, 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_prev, prev_xy, (size_t)n1 * 8, hipMemcpyHostToDevice, ctx->stream));
    WinParams P;
    memset(&P, 0, sizeof P);
    P.frames = ctx->d_frames; P.qsets = ctx->d_qsets;
    P.frame_ids = d_ids; P.qset_ids = d_ids + 1;
    P.prev_xy = d_prev; P.out = d_m12; P.counts = d_count;
    P.th = 1.0f; P.ratio = ratio; P.dist_th = ORBM_TH_LOW; P.check_ori = check_ori; P.stride = n1; P.window = window;
    rc = launch_window(ctx, MODE_INIT, P, 1, std::max(n2, 1));
    if (rc) return rc;
    int32_t count = 0;
    HIPCHK(ctx, hipMemcpyAsync(&count, d_count, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(prev_xy, d_prev, (size_t)n1 * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(matches12, d_m12, (size_t)n1 * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return count;
}

extern "C" int orbm_search_window_free_batch(orbm_ctx *ctx, int n_inst, const int32_t *frame, const int32_t *qset,
                                             int dist_th, int chi2_gate, int32_t stride_q, int32_t *best_idx,
                                             int io_on_device) {
    if (!ctx) return ORBM_EINVAL;
    if (n_inst < 0 || (n_inst > 0 && (!frame || !qset || !best_idx)) || stride_q < 0)
        return orbm_fail(ctx, ORBM_EINVAL, "search_window_free_batch: null argument or negative size");
    if (n_inst == 0 || stride_q == 0) return ORBM_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!io_on_device) {
        int rc = orbm_check_frames(ctx, n_inst, frame, "frame");
        if (!rc) rc = orbm_check_qsets(ctx, n_inst, qset, 1, "qset");
        if (rc) return rc;
        for (int k = 0; k < n_inst; k++) {
            if (ctx->h_qsets[qset[k]].n > stride_q)
                return orbm_fail(ctx, ORBM_EINVAL, "search_window_free_batch: stride_q %d < %d queries of instance %d", stride_q, ctx->h_qsets[qset[k]].n, k);
            if (chi2_gate && ctx->h_qsets[qset[k]].n > 0 && !ctx->h_qsets[qset[k]].f3)
                return orbm_fail(ctx, ORBM_EINVAL, "search_window_free_batch: chi2_gate needs the query set's ur array");
        }
    }
    const size_t out_elems = (size_t)n_inst * stride_q;
    IoStage io(ctx, io_on_device);
    if (!io_on_device) {
        int rc = orbm_ws_reserve(ctx, (size_t)n_inst * 8 + out_elems * 4 + 4096);
        if (rc) return rc;
    }
    const void *dfr, *dqs; void *dout;
    int rc;
    if ((rc = io.in(frame, (size_t)n_inst * 4, &dfr)) || (rc = io.in(qset, (size_t)n_inst * 4, &dqs)) ||
        (rc = io.out(best_idx, out_elems * 4, &dout)))
        return rc;
    HIPCHK(ctx, hipMemsetAsync(dout, 0xFF, out_elems * 4, ctx->stream));   // rows beyond a set's n stay -1
    WinParams P;
    memset(&P, 0, sizeof P);
    P.frames = ctx->d_frames; P.qsets = ctx->d_qsets;
    P.frame_ids = (const int32_t *)dfr; P.qset_ids = (const int32_t *)dqs;
    P.out = (int32_t *)dout; P.dist_th = dist_th; P.stride = stride_q;
    const int chunks = (stride_q + 255) / 256;
    hipLaunchKernelGGL(free_kernel, dim3((unsigned)(n_inst * chunks)), dim3(256), 0, ctx->stream, P, chunks, chi2_gate);
    HIPCHK(ctx, hipGetLastError());
    return io.finish();
}
