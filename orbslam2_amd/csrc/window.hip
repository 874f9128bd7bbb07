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
//              and, where a ratio test reads it, on its second-best free slot whatever its distance;
//   contested: a query is contested if a lower-index unresolved query posted on its best or second-best free slot;
//              a contested query posts on ALL its free candidates (any of them may end up in its top two);
//   decide   : an uncontested query is FINAL if it takes nothing, or if no lower-index unresolved query has its best
//              slot posted; final queries apply their result to the slot state.
// Slots only ever become blocked and a post means "a lower query's outcome may depend on this slot staying free", so
// the best / second-best pair of a final query is the one the sequential loop would have found; the lowest
// unresolved query is always final, so every round makes progress.  Typical inputs need 2-4 rounds per chunk.
#include <limits.h>
#include <string.h>
#include <algorithm>
#include <type_traits>
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

// Read-only frame data the walk and the distance loop touch, either in global memory (typed so: global_load) or
// staged in LDS.
template <bool LDS>
struct FrameAccT {
    std::conditional_t<LDS, const uint32_t *, gptr<const uint32_t>> cell_off;   // 3073
    std::conditional_t<LDS, const v4f *, gptr<const v4f>> rec;                  // cell-ordered {x, y, uright, idx | oct << 16}
    std::conditional_t<LDS, const v4u *, gptr<const v4u>> desc;                 // 2 per keypoint
    float min_x, min_y, inv_w, inv_h;
};

// Frame::GetFeaturesInArea (Frame.cc:434-498) / KeyFrame::GetFeaturesInArea (KeyFrame.cc:766-812):
// calls body(k, octave, uright) for every keypoint the reference would push, in the reference's order
// (ix ascending, iy ascending, insertion order).  Cells (ix, minY..maxY) are consecutive in the CSR, so a
// column of the window is ONE contiguous run of 16-byte records.
template <bool KF_MODE, typename Acc, typename Body>
__device__ __forceinline__ void walk_window(const Acc &f, float x, float y, float r, int minLevel, int maxLevel,
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
            const v4f rc = f.rec[j];
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

constexpr int SCAN_U = 8;      // split form: queries scanned for liveness per lane and pass
constexpr int WIN_CACHE = 16;  // candidates remembered per query (registers); wider windows fall back to re-walking

struct Query {
    v4u d0, d1;
    float x, y, r, xr, angle;
    int minL, maxL, ql;
    uint32_t flags;
    bool valid;
};

// where an instance's queries come from: a query set, or (MODE_INIT) the keypoints of frame 1
struct QSrc {
    QSetG QS;
    FrameG F1;
    int Q;
};

template <int MODE>
__device__ __forceinline__ QSrc query_source(const WinParams &P, int inst) {
    QSrc S;
    if (MODE == MODE_INIT) { S.F1 = FrameG(P.frames[P.qset_ids[inst]]); S.Q = S.F1.n; }
    else { S.QS = QSetG(P.qsets[P.qset_ids[inst]]); S.Q = S.QS.n; }
    return S;
}

// everything of query qi except its descriptor (F = the frame searched)
template <int MODE>
__device__ __forceinline__ Query load_params(const WinParams &P, const FrameG &F, const QSrc &S, int qi) {
    Query q;
    q.d0 = q.d1 = v4u{0, 0, 0, 0};
    q.x = q.y = q.r = q.xr = q.angle = 0.f; q.minL = q.maxL = -1; q.ql = 0; q.flags = 0; q.valid = false;
    if (qi >= S.Q) return q;
    if (MODE == MODE_INIT) {
        if (S.F1.octave[qi] == 0) {                                                  // :582-584
            q.valid = true;
            q.x = P.prev_xy[2 * qi]; q.y = P.prev_xy[2 * qi + 1];
            q.r = (float)P.window; q.minL = 0; q.maxL = 0;                            // :587
            q.angle = S.F1.angle[qi];
        }
    } else {
        const QSetG &QS = S.QS;
        q.flags = QS.flags[qi];
        if (q.flags & ORBM_Q_VALID) {
            q.valid = true;
            q.x = QS.f0[qi]; q.y = QS.f1[qi];
            if (MODE == MODE_PROJ) {
                const int lvl = QS.l0[qi];
                float rv = ((double)QS.f3[qi] > 0.998) ? 2.5f : 4.0f;              // :168-174
                if (P.th != 1.0f) rv = __fmul_rn(rv, P.th);                          // :61, :84-85
                q.r = __fmul_rn(rv, F.sf[lvl]);                                      // :89
                q.minL = lvl - 1; q.maxL = lvl;
                q.xr = QS.f2[qi];
            } else {
                q.r = QS.f2[qi]; q.minL = QS.l0[qi]; q.maxL = QS.l1[qi];
                if (QS.f3) q.xr = QS.f3[qi];
                if (QS.angle) q.angle = QS.angle[qi];
            }
        }
    }
    return q;
}

template <int MODE>
__device__ __forceinline__ void load_desc(const QSrc &S, int qi, Query &q) {
    const gptr<const v4u> p4 = (gptr<const v4u>)(MODE == MODE_INIT ? S.F1.desc : S.QS.desc) + 2 * (size_t)qi;
    q.d0 = p4[0]; q.d1 = p4[1];
}

// the gates of the reference's inner loop that do not depend on the slot state
template <int MODE>
__device__ __forceinline__ bool static_gates(const Query &q, bool stereo_gate, int oct, float ur) {
    if (MODE == MODE_SIM3) { if (oct < q.minL || oct > q.maxL) return false; }        // :515-518
    if (stereo_gate) {
        if (ur > 0) {
            const float er = fabsf(__fsub_rn(q.xr, ur));
            if (er > q.r) return false;                                                // :117-122 / :1883-1891
        }
    }
    return true;
}

template <int MODE>
__device__ __forceinline__ bool has_stereo_gate(const QSrc &S) {
    return MODE == MODE_PROJ || (MODE == MODE_LAST && S.QS.f3 != nullptr);
}

// ---- split form, kernel 1 of 2: ONE window walk per query, anywhere on the chip.  Nothing here depends on the slot
// state, so queries are independent.  A workgroup stages the frame's grid and cell-ordered records in LDS once and
// its waves then take rows of 64 queries (one per lane) with no barrier between them.  Per lane: a single flat loop
// over the window's records (column changes are just another iteration, so a wave iterates max-over-lanes of the
// window's size, not the sum over columns of the widest column run), branch-free gates, survivors parked in LDS;
// then a second, wave-uniform loop computes the Hamming distances.  Each candidate that passes the static gates is
// stored as k | oct << 16 | dist << 20 in reference visit order, in rows of 64 lanes (row i of a wave = candidate i
// of its 64 queries; only rows some lane needs are written).
constexpr int COLLECT_THREADS = 1024;
constexpr int COST_CLASSES = 64;

template <int MODE>
__global__ __launch_bounds__(COLLECT_THREADS) void collect_kernel(WinParams P, int waves_per_inst, int wgs_per_inst,
                                                                  uint32_t *cand, uint32_t *cand_hi, uint8_t *ncand_out,
                                                                  uint64_t *postable_out, int cap, int stage_nxt, int groups) {
    extern __shared__ __align__(16) uint32_t lds[];
    const int inst = blockIdx.x / wgs_per_inst, part = blockIdx.x - inst * wgs_per_inst;
    const int tid = threadIdx.x;
    const FrameG F(P.frames[P.frame_ids[inst]]);
    const int n = F.n;
    uint32_t *l_cell = lds;                                                      // 3076
    v4f *l_rec = reinterpret_cast<v4f *>(lds + 3076);                            // 4n
    // skip lists (orbm_internal.h, cell_nxt): staged when the frame has them and the launch left room
    const bool have_nxt = stage_nxt && F.cell_nxt != nullptr;
    uint16_t *l_nxt = reinterpret_cast<uint16_t *>(lds + 3076 + 4 * (size_t)n);  // (n + 1) x 8
    uint16_t *l_cc = l_nxt + (stage_nxt ? 8 * ((size_t)n + 1) : 0);              // WIN_CACHE x COLLECT_THREADS stream positions
    int *l_bucket = (int *)(l_cc + WIN_CACHE * COLLECT_THREADS);                 // cost classes
    uint16_t *l_perm = reinterpret_cast<uint16_t *>(l_bucket + COST_CLASSES);    // 2 x COLLECT_THREADS
    for (int i = tid; i <= ORBM_GRID_CELLS; i += COLLECT_THREADS) l_cell[i] = F.cell_off[i];
    for (int i = tid; i < n; i += COLLECT_THREADS) l_rec[i] = F.cell_rec[i];
    if (have_nxt) {
        const gptr<const v4u> g = (gptr<const v4u>)F.cell_nxt;
        v4u *l = reinterpret_cast<v4u *>(l_nxt);
        const int rows = (int)F.cell_off[ORBM_GRID_CELLS] + 1;
        for (int i = tid; i < rows; i += COLLECT_THREADS) l[i] = g[i];
    }
    __syncthreads();
    const QSrc S = query_source<MODE>(P, inst);
    const bool stereo_gate = has_stereo_gate<MODE>(S);
    const gptr<const v4u> gdesc = (gptr<const v4u>)F.desc;
    const float min_x = F.min_x, min_y = F.min_y, inv_w = F.inv_w, inv_h = F.inv_h;
    constexpr int WAVES = COLLECT_THREADS / 64;
    // queries sorted together: `groups` rows of 64 per wave (2; 1 when the launch has few instances, so that one instance
    // spreads over more workgroups)
    const int SUPER = groups * COLLECT_THREADS;
    const int lane = tid & 63, wave = tid >> 6;
    // A wave costs what its widest window costs and window areas differ by more than 10x between pyramid levels, so
    // the queries of a super-chunk are counting-sorted by the length of their walk (columns + records) and each
    // wave takes one heavy and one light group of 64 (results are still written at the query's own position).
    const int n_super = (waves_per_inst * 64 + SUPER - 1) / SUPER;
    for (int sc = part; sc < n_super; sc += wgs_per_inst) {
        if (tid < COST_CLASSES) l_bucket[tid] = 0;
        __syncthreads();
        int cls[2], rank[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (h >= groups) { cls[h] = 0; rank[h] = 0; continue; }
            const Query t = load_params<MODE>(P, F, S, sc * SUPER + h * COLLECT_THREADS + tid);
            int c = 0;
            if (t.valid) {
                const float xr = __fsub_rn(t.x, min_x), yr = __fsub_rn(t.y, min_y);
                const int x0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(xr, t.r), inv_w)));
                const int x1 = min(ORBM_GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(xr, t.r), inv_w)));
                const int y0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(yr, t.r), inv_h)));
                const int y1 = min(ORBM_GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(yr, t.r), inv_h)));
                if (x1 >= x0 && y1 >= y0) {
                    int recs = 0;                              // loop iterations of the walk: columns + records visited
                    for (int ix = x0; ix <= x1; ix++)
                        recs += (int)(l_cell[ix * ORBM_GRID_ROWS + y1 + 1] - l_cell[ix * ORBM_GRID_ROWS + y0]);
                    const bool hop = have_nxt && (MODE == MODE_PROJ || MODE == MODE_SIM3 || MODE == MODE_INIT ||
                                                  (t.maxL >= 0 && t.maxL <= 7 && t.maxL - t.minL <= 1));
                    const int cnt = x1 - x0 + 1 + (hop ? recs / 3 : recs);
                    c = cnt < 62 ? 1 + (cnt >> 1) : min(COST_CLASSES - 1, 32 + ((cnt - 62) >> 4));
                }
            }
            cls[h] = c;
            rank[h] = atomicAdd(&l_bucket[c], 1);
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (h >= groups) continue;
            int start = 0;
            for (int bk = COST_CLASSES - 1; bk > cls[h]; bk--) start += l_bucket[bk];
            l_perm[start + rank[h]] = (uint16_t)(h * COLLECT_THREADS + tid);
        }
        __syncthreads();
      for (int h = 0; h < groups; h++) {
        const int group = h == 0 ? wave : 2 * WAVES - 1 - wave;
        const int qi = sc * SUPER + (int)l_perm[group * 64 + lane];
        const int rowi = qi >> 6;
        Query q = load_params<MODE>(P, F, S, qi);
        if (q.valid) load_desc<MODE>(S, qi, q);
        // Frame.cc:434-498 / KeyFrame.cc:766-812, as walk_window()
        const float xr = __fsub_rn(q.x, min_x), yr = __fsub_rn(q.y, min_y), r = q.r;
        const int x0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(xr, r), inv_w)));
        const int x1 = min(ORBM_GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(xr, r), inv_w)));
        const int y0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(yr, r), inv_h)));
        const int y1 = min(ORBM_GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(yr, r), inv_h)));
        // level gates: Frame's GetFeaturesInArea filters when minLevel > 0 or maxLevel >= 0 (a negative maxLevel
        // leaves the top open); KeyFrame's does not filter, SearchByProjection(KF, Scw, ...) does it itself (:515-518)
        int lo = -1, hi = 255;
        if (MODE == MODE_SIM3) { lo = q.minL; hi = q.maxL; }
        else if (q.minL > 0 || q.maxL >= 0) { lo = q.minL; if (q.maxL >= 0) hi = q.maxL; }
        const uint32_t qd[8] = {q.d0.x, q.d0.y, q.d0.z, q.d0.w, q.d1.x, q.d1.y, q.d1.z, q.d1.w};
        // candidates 0..WIN_CACHE-1 of a query live in `cand`, the rest (up to cap) in `cand_hi`
        uint32_t *row = cand + ((size_t)inst * waves_per_inst + rowi) * (WIN_CACHE * 64) + (qi & 63);
        uint32_t *row_hi = cand_hi + ((size_t)inst * waves_per_inst + rowi) * ((size_t)(cap - WIN_CACHE) * 64) + (qi & 63);
        // claimable candidates (distance <= the acceptance threshold): up to three slot ids, their count in bits 48-49,
        // bit 50 = "treat as live" (more than three, or more candidates than a row holds)
        uint64_t pw = 0;
        int npost = 0;
        int ncand = 0;      // candidates that passed the static gates
        auto finish = [&](uint32_t v) {                         // v = k | oct << 16 -> adds the distance, notes claimables
            const int k = (int)(v & 0xFFFFu);
            const v4u dlo = gdesc[2 * k], dhi = gdesc[2 * k + 1];
            const uint32_t cd[8] = {dlo.x, dlo.y, dlo.z, dlo.w, dhi.x, dhi.y, dhi.z, dhi.w};
            const uint32_t dist = hamming256(qd, cd);
            if ((int)dist <= P.dist_th) {
                if (npost < 3) pw |= (uint64_t)(uint32_t)k << (16 * npost);
                npost++;
            }
            return v | (dist << 20);
        };
        // first phase: a single flat loop over the window's records (a column change is just another iteration);
        // survivors are parked (LDS for the first WIN_CACHE, the row itself beyond) until the second phase
        if (q.valid && x0 < ORBM_GRID_COLS && x1 >= 0 && y0 < ORBM_GRID_ROWS && y1 >= 0) {
            int ix = x0 - 1;
            uint32_t j = 0, e = 0;
            bool more = true;
            // octaves {hi-1, hi} cover [lo, hi] when hi - lo <= 1: hop along that pair's skip list (the gates below
            // stay, the list only leaves out records that cannot pass the level gate)
            const bool hop = have_nxt && hi <= 7 && hi - lo <= 1;
            const uint16_t *nx = l_nxt + (hop ? hi : 0);
            while (more) {
                if (j < e) {
                    const uint32_t jcur = j;
                    const v4f rc = l_rec[j];
                    j = hop ? (uint32_t)nx[(j + 1) * 8] : j + 1;
                    const uint32_t w = __float_as_uint(rc.w);
                    const int oct = (int)(w >> 16);
                    int pass = (int)(oct >= lo) & (int)(oct <= hi) & (int)(fabsf(__fsub_rn(rc.x, q.x)) < r) & (int)(fabsf(__fsub_rn(rc.y, q.y)) < r);
                    if (stereo_gate) pass &= 1 ^ ((int)(rc.z > 0) & (int)(fabsf(__fsub_rn(q.xr, rc.z)) > r));   // :117-122 / :1883-1891
                    if (pass) {
                        if (ncand < WIN_CACHE) l_cc[ncand * COLLECT_THREADS + tid] = (uint16_t)jcur;
                        else if (ncand < cap) row_hi[(ncand - WIN_CACHE) * 64] = w;
                        ncand++;
                    }
                }
                if (j >= e) {                                  // run finished (or empty): the next column, in the same iteration
                    if (++ix > x1) more = false;
                    else {
                        j = l_cell[ix * ORBM_GRID_ROWS + y0];
                        e = l_cell[ix * ORBM_GRID_ROWS + y1 + 1];
                        if (hop) j = nx[j * 8];
                    }
                }
                // keeps this ONE loop: without a convergent operation here the compiler splits it into nested
                // "advance until a survivor" loops, and lanes then wait for each other at every survivor
                __builtin_amdgcn_wave_barrier();
            }
        }
        // second phase: Hamming distances, wave-uniform trip counts
        {
            int wn = min(ncand, cap);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) wn = max(wn, __shfl_xor(wn, off, 64));
            wn = __builtin_amdgcn_readfirstlane(wn);
            for (int i = 0; i < min(wn, WIN_CACHE); i++)
                if (i < ncand) row[i * 64] = finish(__float_as_uint(l_rec[l_cc[i * COLLECT_THREADS + tid]].w));
            for (int i = WIN_CACHE; i < wn; i++)
                if (i < ncand) row_hi[(i - WIN_CACHE) * 64] = finish(row_hi[(i - WIN_CACHE) * 64]);
        }
        pw |= (uint64_t)min(npost, 3) << 48;
        if (npost > 3 || ncand > cap) pw |= 1ull << 50;
        if (qi < waves_per_inst * 64) {
            ncand_out[(size_t)inst * waves_per_inst * 64 + qi] = (uint8_t)min(ncand, 255);
            postable_out[(size_t)inst * waves_per_inst * 64 + qi] = pw;
        }
      }
        __syncthreads();     // l_perm and l_bucket are rewritten by the next super-chunk
    }
}

// ---- the claim loop.  PRE = false: collects inline (one kernel does everything; frame optionally staged in LDS).
// PRE = true: kernel 2 of 2 of the split form, candidates come from collect_kernel's rows.
template <int MODE, bool LDS_FRAME, bool PRE>
__global__ void window_kernel(WinParams P, int waves_per_inst, const uint32_t *cand, const uint8_t *ncand_in,
                              const uint64_t *postable_in, int cap, const uint32_t *cand_hi) {
    extern __shared__ __align__(16) uint32_t lds[];
    const int inst = blockIdx.x;
    const FrameG F(P.frames[P.frame_ids[inst]]);
    const int n = F.n;
    const int tid = threadIdx.x, nthr = blockDim.x;
    // LDS carve.  LDS_FRAME: [cell_off 3076][rec 4n][desc 8n] then the slot state; else slot state only.
    FrameAccT<LDS_FRAME> acc;
    acc.min_x = F.min_x; acc.min_y = F.min_y; acc.inv_w = F.inv_w; acc.inv_h = F.inv_h;
    uint32_t *base = lds;
    if constexpr (LDS_FRAME) {
        uint32_t *l_cell = lds;
        v4f *l_rec = reinterpret_cast<v4f *>(lds + 3076);
        v4u *l_desc = reinterpret_cast<v4u *>(lds + 3076 + 4 * (size_t)n);
        for (int i = tid; i <= ORBM_GRID_CELLS; i += nthr) l_cell[i] = F.cell_off[i];
        for (int i = tid; i < n; i += nthr) l_rec[i] = F.cell_rec[i];
        const gptr<const v4u> gd = (gptr<const v4u>)F.desc;
        for (int i = tid; i < 2 * n; i += nthr) l_desc[i] = gd[i];
        acc.cell_off = l_cell; acc.rec = l_rec; acc.desc = l_desc;
        base = lds + 3076 + 12 * (size_t)n;
    } else {
        acc.cell_off = F.cell_off; acc.rec = F.cell_rec; acc.desc = (gptr<const v4u>)F.desc;
    }
    uint32_t *state = base;                       // ORBM_MP_* flags | vMatchedDistance
    int32_t *owner = (int32_t *)(base + n);
    uint32_t *post0 = base + 2 * (size_t)n, *post1 = base + 3 * (size_t)n;
    uint32_t *binmask = base + 4 * (size_t)n;
    int *hist = (int *)(base + 5 * (size_t)n);
    int *nmatch = hist + 32;
    int *keep = hist + 33;

    const QSrc S = query_source<MODE>(P, inst);
    const int Q = S.Q;
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
                          (MODE == MODE_INIT || S.QS.angle != nullptr);
    const bool stereo_gate = has_stereo_gate<MODE>(S);
    const int thr = MODE == MODE_PROJ ? ORBM_TH_HIGH : (MODE == MODE_INIT ? ORBM_TH_LOW : P.dist_th);
    int my_matches = 0;
    const uint8_t *nc_in = PRE ? ncand_in + (size_t)inst * waves_per_inst * 64 : nullptr;
    const uint32_t *rows = PRE ? cand + (size_t)inst * waves_per_inst * (WIN_CACHE * 64) : nullptr;
    const uint32_t *rows_hi = PRE ? cand_hi + (size_t)inst * waves_per_inst * ((size_t)(cap - WIN_CACHE) * 64) : nullptr;

    auto load_nc = [&](int qi) { return (PRE && qi < Q) ? (int)nc_in[qi] : 0; };
    auto load_query = [&](int qi, int nc) {
        Query q = load_params<MODE>(P, F, S, qi);
        if (PRE) {
            if (nc > cap) load_desc<MODE>(S, qi, q);           // only the re-walking fallback compares descriptors here
        } else if (q.valid) {
            load_desc<MODE>(S, qi, q);
        }
        return q;
    };
    auto wave_max = [&](int v) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
        return __builtin_amdgcn_readfirstlane(v);
    };
    auto is_blocked = [&](uint32_t st, int dist) {
        if (MODE == MODE_PROJ || MODE == MODE_LAST) return (st & ORBM_MP_PRESENT) && (st & ORBM_MP_OBS);   // :113-115 / :1879-1881
        if (MODE == MODE_INIT) return (int)st <= dist;                                                     // :608
        return (st & ORBM_MP_PRESENT) != 0;                                                                // :2048 / :512
    };

    // One chunk of up to blockDim queries, one per lane, in ascending query index over tid (claim priority = tid).
    // Called by every thread of the workgroup (barriers inside).
    auto process_chunk = [&](const int qi, const Query &cur, const int nc_pre, uint32_t (&cc)[WIN_CACHE]) {
        bool unresolved = cur.valid;
        const uint32_t qd[8] = {cur.d0.x, cur.d0.y, cur.d0.z, cur.d0.w, cur.d1.x, cur.d1.y, cur.d1.z, cur.d1.w};
        const float x = cur.x, y = cur.y, r = cur.r, qangle = cur.angle;
        const int minL = cur.minL, maxL = cur.maxL;
        const uint32_t qflags = cur.flags;

        // ---- collect: ONE window walk per query.  The static gates (level range, |dx|<r, |dy|<r, octave filter,
        // stereo gate) and the Hamming distance do not depend on the slot state, so the candidates that pass are
        // remembered as k | oct << 16 | dist << 20 and the rounds below never walk again.  More than WIN_CACHE
        // candidates (wide windows): the rounds fall back to walking.
        int ncand = 0;
        if (PRE) {
            ncand = nc_pre;
        } else if (unresolved) {
            auto collect = [&](int k, int oct, float ur) {
                if (!static_gates<MODE>(cur, stereo_gate, oct, ur)) return;
                if (ncand < WIN_CACHE) {
                    const v4u lo = acc.desc[2 * k], hi = acc.desc[2 * k + 1];
                    const uint32_t cd[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    const uint32_t v = (uint32_t)k | ((uint32_t)oct << 16) | (hamming256(qd, cd) << 20);
#pragma unroll
                    for (int i = 0; i < WIN_CACHE; i++) if (i == ncand) cc[i] = v;
                }
                ncand++;
            };
            if (MODE == MODE_SIM3) walk_window<true>(acc, x, y, r, minL, maxL, collect);
            else walk_window<false>(acc, x, y, r, minL, maxL, collect);
        }
        if (ncand == 0) unresolved = false;   // empty window: nothing to match, nothing to claim (:91-92)
        // candidate i of this lane's query: the first WIN_CACHE in registers, the rest (split form, rows of `cap`
        // entries) re-read from the row; i is wave-uniform
        const bool cached = ncand <= cap;
        const int wave_ncand = wave_max(min(ncand, cap));
        const uint32_t *myrow_hi = PRE ? rows_hi + (size_t)(qi >> 6) * ((size_t)(cap - WIN_CACHE) * 64) + (qi & 63) : nullptr;
        auto cand_at = [&](int i) -> uint32_t {
            if (i < WIN_CACHE) return cc[i];
            return (PRE && i < ncand) ? myrow_hi[(i - WIN_CACHE) * 64] : 0u;
        };

        // A query none of whose claimable candidates (distance <= threshold) is still free can never be accepted, and
        // blocked slots stay blocked, so it is settled without taking part in the rounds; a chunk with no live query
        // costs one barrier.
        if (cached && unresolved) {
            bool live = false;
            for (int i = 0; i < wave_ncand; i++) {
                const uint32_t v = cand_at(i);
                if (i < ncand && (int)(v >> 20) <= thr) {
                    const uint32_t st = state[v & 0xFFFFu];
                    bool blocked;
                    if (MODE == MODE_PROJ || MODE == MODE_LAST) blocked = (st & ORBM_MP_PRESENT) && (st & ORBM_MP_OBS);
                    else if (MODE == MODE_INIT) blocked = (int)st <= (int)(v >> 20);
                    else blocked = (st & ORBM_MP_PRESENT) != 0;
                    if (!blocked) live = true;
                }
            }
            unresolved = live;
        }
        if (!__syncthreads_or(unresolved ? 1 : 0)) return;

        int round = 0;
        while (true) {
            uint32_t *post = (round & 1) ? post1 : post0;
            uint32_t *post_next = (round & 1) ? post0 : post1;
            int bestDist = MODE == MODE_INIT ? INT_MAX : 256, bestDist2 = bestDist, bestIdx = -1, bestIdx2 = -1;
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
                if (dist <= thr) atomicMin(&post[k], (uint32_t)tid);                       // this query may still claim k
                // strict '<' on the best, else strict '<' on the second (:124-137 etc.), as selects: with branches the
                // compiler keeps the two indices in a scratch-memory array and stores through a computed offset
                const bool b1 = dist < bestDist, b2 = !b1 && dist < bestDist2;
                bestDist2 = b1 ? bestDist : (b2 ? dist : bestDist2);
                bestLevel2 = b1 ? bestLevel : (b2 ? oct : bestLevel2);
                bestIdx2 = b1 ? bestIdx : (b2 ? k : bestIdx2);
                bestDist = b1 ? dist : bestDist;
                bestLevel = b1 ? oct : bestLevel;
                bestIdx = b1 ? k : bestIdx;
            };
            // ---- evaluate ---------------------------------------------------------------------------
            if (unresolved) {
                if (cached) {
                    // wave-uniform trip count: the register array is indexed by a scalar (v_movrels), no unrolling
                    for (int i = 0; i < wave_ncand; i++) {
                        const uint32_t v = cand_at(i);
                        if (i < ncand) consider((int)(v & 0xFFFFu), (int)((v >> 16) & 15u), (int)(v >> 20));
                    }
                } else {
                    auto body = [&](int k, int oct, float ur) {
                        if (!static_gates<MODE>(cur, stereo_gate, oct, ur)) return;
                        const v4u lo = acc.desc[2 * k], hi = acc.desc[2 * k + 1];
                        const uint32_t cd[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                        consider(k, oct, (int)hamming256(qd, cd));
                    };
                    if (MODE == MODE_SIM3) walk_window<true>(acc, x, y, r, minL, maxL, body);
                    else walk_window<false>(acc, x, y, r, minL, maxL, body);
                }
            }
            // a query that can still be accepted also names its second-best free slot where a ratio test reads it: lower
            // queries' interest in a slot must be visible to whoever wants to take it (claimable slots are posted above)
            constexpr bool USES_SECOND = MODE == MODE_PROJ || MODE == MODE_INIT;
            const bool live = unresolved && bestIdx >= 0 && bestDist <= thr;
            if (live && USES_SECOND && bestIdx2 >= 0) atomicMin(&post[bestIdx2], (uint32_t)tid);
            __syncthreads();
            // ---- contested? ---------------------------------------------------------------------------
            // The outcome is a function of the best and (where a ratio test reads it) the second-best free candidate, and
            // slots only ever become blocked.  If a lower-index unresolved query may still take one of those two, this
            // query is not settled, and ANY of its free candidates may end up in its top two: it posts on all of them so
            // that no later query takes one of them first.
            bool contested = false;
            if (live) {
                contested = post[bestIdx] < (uint32_t)tid;
                if (USES_SECOND && bestIdx2 >= 0 && post[bestIdx2] < (uint32_t)tid) contested = true;
            }
            if (__syncthreads_or(contested ? 1 : 0)) {
                if (contested) {
                    if (cached) {
                        for (int i = 0; i < wave_ncand; i++) {
                            const uint32_t v = cand_at(i);
                            if (i < ncand) {
                                const int k = (int)(v & 0xFFFFu);
                                if (!is_blocked(state[k], (int)(v >> 20))) atomicMin(&post[k], (uint32_t)tid);
                            }
                        }
                    } else {
                        auto body = [&](int k, int oct, float ur) {
                            if (!static_gates<MODE>(cur, stereo_gate, oct, ur)) return;
                            if (MODE == MODE_INIT) {
                                const v4u lo = acc.desc[2 * k], hi = acc.desc[2 * k + 1];
                                const uint32_t cd[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                                if (is_blocked(state[k], (int)hamming256(qd, cd))) return;
                            } else if (is_blocked(state[k], 0)) return;
                            atomicMin(&post[k], (uint32_t)tid);
                        };
                        if (MODE == MODE_SIM3) walk_window<true>(acc, x, y, r, minL, maxL, body);
                        else walk_window<false>(acc, x, y, r, minL, maxL, body);
                    }
                }
                __syncthreads();
            }
            // ---- decide -----------------------------------------------------------------------------
            if (unresolved) {
                // not live: a best distance above the threshold can only get worse, final at once.  Otherwise final when
                // uncontested and, if it takes its best slot, no lower-index unresolved query has that slot posted.
                bool is_final = !live;
                if (live && !contested) {
                    bool accept0;
                    if (MODE == MODE_PROJ) accept0 = !(bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(P.ratio, (float)bestDist2));
                    else if (MODE == MODE_INIT) accept0 = (float)bestDist < __fmul_rn((float)bestDist2, P.ratio);
                    else accept0 = true;
                    is_final = !accept0 || post[bestIdx] >= (uint32_t)tid;
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

    };

    if (!PRE) {
        // single-kernel form: queries in index order, the NEXT chunk's query is loaded while the current one is matched
        Query nextq = load_query(tid, 0);
        for (int base_q = 0; base_q < Q; base_q += nthr) {
            const int qi = base_q + tid;
            const Query cur = nextq;
            nextq = load_query(qi + nthr, 0);
            uint32_t cc[WIN_CACHE];
#pragma unroll
            for (int i = 0; i < WIN_CACHE; i++) cc[i] = 0;
            process_chunk(qi, cur, 0, cc);
        }
    } else {
        // split form.  collect_kernel left one word per query naming its claimable candidates (distance <= threshold).
        // A query none of whose claimable candidates is still free can never be accepted, and blocked slots stay
        // blocked, so the queries are first scanned SCAN_U x blockDim at a time against the current slot state (a few
        // instructions each, loads independent) and only the live ones, compacted in index order, go through the
        // rounds.  Typical inputs leave a few percent of the queries live.
        uint16_t *qlist = reinterpret_cast<uint16_t *>(hist + 64);          // SCAN_U x blockDim
        int *cnts = hist + 64 + SCAN_U * nthr / 2;                          // SCAN_U x waves (+1)
        int *pre = cnts + SCAN_U * 16 + 1;
        const int wave = tid >> 6, lane = tid & 63, NW = nthr >> 6;
        const uint64_t *pw = postable_in + (size_t)inst * waves_per_inst * 64;
        for (int sb = 0; sb < Q; sb += SCAN_U * nthr) {
            uint64_t w[SCAN_U];
#pragma unroll
            for (int u = 0; u < SCAN_U; u++) {
                const int qi = sb + u * nthr + tid;
                w[u] = qi < Q ? pw[qi] : 0ull;
            }
            uint32_t lm = 0, ranks = 0, rankhi = 0;
#pragma unroll
            for (int u = 0; u < SCAN_U; u++) {
                bool live = (w[u] >> 50) & 1ull;
                const int c = (int)((w[u] >> 48) & 3ull);
#pragma unroll
                for (int j = 0; j < 3; j++)
                    if (j < c && !is_blocked(state[(uint32_t)(w[u] >> (16 * j)) & 0xFFFFu], 0)) live = true;
                const uint64_t m = __ballot(live);
                if (lane == 0) cnts[u * NW + wave] = __popcll(m);
                const uint32_t rk = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (u < 4) ranks |= rk << (8 * u); else rankhi |= rk << (8 * (u - 4));
                if (live) lm |= 1u << u;
            }
            __syncthreads();
            if (tid < SCAN_U * NW) {
                int sum = 0;
                for (int e2 = 0; e2 < tid; e2++) sum += cnts[e2];
                pre[tid] = sum;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < SCAN_U; u++)
                if ((lm >> u) & 1u) {
                    const uint32_t rk = u < 4 ? (ranks >> (8 * u)) & 0xFFu : (rankhi >> (8 * (u - 4))) & 0xFFu;
                    qlist[pre[u * NW + wave] + rk] = (uint16_t)(u * nthr + tid);
                }
            const int nlive = pre[SCAN_U * NW - 1] + cnts[SCAN_U * NW - 1];
            __syncthreads();
            for (int c0 = 0; c0 < nlive; c0 += nthr) {
                const int li = c0 + tid;
                const int qi = li < nlive ? sb + (int)qlist[li] : Q;
                const int nc = load_nc(qi);
                const Query cur = load_query(qi, nc);
                uint32_t cc[WIN_CACHE];
                const uint32_t *row = rows + (size_t)(qi >> 6) * (WIN_CACHE * 64) + (qi & 63);
#pragma unroll
                for (int i = 0; i < WIN_CACHE; i++) cc[i] = i < min(nc, WIN_CACHE) ? row[i * 64] : 0u;
                process_chunk(qi, cur, nc, cc);
            }
        }
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
                const int bin = rot_bin(S.F1.angle[i], F.angle[m]);
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
    const FrameG F(P.frames[P.frame_ids[inst]]);
    const QSetG QS(P.qsets[P.qset_ids[inst]]);
    const int qi = chunk * 256 + (int)threadIdx.x;
    if (qi >= QS.n || qi >= P.stride) return;
    int32_t *out = P.out + (size_t)inst * P.stride;
    int result = -1;
    if (QS.flags[qi] & ORBM_Q_VALID) {
        FrameAccT<false> acc;
        acc.min_x = F.min_x; acc.min_y = F.min_y; acc.inv_w = F.inv_w; acc.inv_h = F.inv_h;
        acc.cell_off = F.cell_off; acc.rec = F.cell_rec; acc.desc = (gptr<const v4u>)F.desc;
        const float u = QS.f0[qi], v = QS.f1[qi], radius = QS.f2[qi];
        const float ur = QS.f3 ? QS.f3[qi] : 0.f;
        const int minL = QS.l0[qi], maxL = QS.l1[qi];
        const gptr<const v4u> p4 = (gptr<const v4u>)QS.desc + 2 * (size_t)qi;
        const v4u d0 = p4[0], d1 = p4[1];
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
            const v4u lo = acc.desc[2 * k], hi = acc.desc[2 * k + 1];
            const uint32_t cd[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            const int dist = (int)hamming256(qd, cd);
            if (dist < bestDist) { bestDist = dist; bestIdx = k; }
        };
        walk_window<true>(acc, u, v, radius, -1, -1, body);
        if (bestDist <= P.dist_th) result = bestIdx;
    }
    out[qi] = result;
}

constexpr int WIN_LDS_FRAME_MAX_N = 2200;   // cell_off + 16 B records + 32 B descriptors + 20 B state per keypoint <= 160 KB

template <int MODE, bool LDS_FRAME, bool PRE>
int launch_claims(orbm_ctx *ctx, const WinParams &P, int n_inst, int max_n, int threads, int waves_per_inst,
                  const uint32_t *cand, const uint8_t *ncand, const uint64_t *postable, int cap,
                  const uint32_t *cand_hi = nullptr) {
    const size_t lds = ((size_t)(LDS_FRAME ? 3076 + 12 * (size_t)max_n : 0) + 5 * (size_t)max_n + 64 +
                        (PRE ? SCAN_U * threads / 2 + 2 * (SCAN_U * 16 + 1) : 0)) * 4;
    HIPCHK(ctx, (orbm_allow_dynamic_lds<&window_kernel<MODE, LDS_FRAME, PRE>>(ctx, lds)));
    hipLaunchKernelGGL((window_kernel<MODE, LDS_FRAME, PRE>), dim3(n_inst), dim3(threads), lds, ctx->stream, P,
                       waves_per_inst, cand, ncand, postable, cap, cand_hi);
    HIPCHK(ctx, hipGetLastError());
    return ORBM_OK;
}

// dynamic LDS of collect_kernel: grid, records, optional skip lists, parking (u16), cost classes, lane assignment (u16)
static size_t collect_lds_bytes(int max_n, bool with_nxt) {
    return (3076 + 4 * (size_t)max_n) * 4 + (with_nxt ? 16 * ((size_t)max_n + 1) : 0) + (size_t)WIN_CACHE * COLLECT_THREADS * 2 +
           COST_CLASSES * 4 + 2 * COLLECT_THREADS * 2 + 64;
}

// tuning / test knob: 0 = split form (collect_kernel + claims), 1 = single kernel with the frame staged in LDS,
// 2 = single kernel reading the frame from global memory
int g_win_variant = 0;

template <int MODE>
int launch_mode(orbm_ctx *ctx, const WinParams &P, int n_inst, int max_n, int max_q) {
    // frames too large for collect_kernel's LDS staging use the single-kernel form
    const bool single = g_win_variant != 0 ||
                        collect_lds_bytes(max_n, false) > 160 * 1024;
    if (single) {
        if (max_n <= WIN_LDS_FRAME_MAX_N && g_win_variant != 2)
            return launch_claims<MODE, true, false>(ctx, P, n_inst, max_n, MODE == MODE_INIT ? 256 : 1024, 0, nullptr, nullptr, nullptr, WIN_CACHE);
        return launch_claims<MODE, false, false>(ctx, P, n_inst, max_n, MODE == MODE_INIT ? 256 : 512, 0, nullptr, nullptr, nullptr, WIN_CACHE);
    }
    {
        const int waves_per_inst = (max_q + 63) / 64;
        if (waves_per_inst == 0) return launch_claims<MODE, false, false>(ctx, P, n_inst, max_n, 256, 0, nullptr, nullptr, nullptr, WIN_CACHE);
        const size_t rows = (size_t)n_inst * waves_per_inst;
        // candidates kept per query: 64 unless the batch is so large that the rows would not fit 8 GB (queries with
        // more candidates than that are still exact: the claims kernel re-walks their window)
        int cap = 64;
        while (cap > WIN_CACHE && rows * cap * 256 > ((size_t)8 << 30)) cap >>= 1;
        const size_t cand_bytes = rows * cap * 64 * 4;
        int rc = orbm_ws2_reserve(ctx, cand_bytes + rows * 64 * 9 + 256);
        if (rc) return rc;
        uint32_t *cand = (uint32_t *)ctx->ws2;                                   // rows x WIN_CACHE x 64
        uint32_t *cand_hi = cand + rows * WIN_CACHE * 64;                        // rows x (cap - WIN_CACHE) x 64
        uint64_t *postable = (uint64_t *)((char *)ctx->ws2 + cand_bytes);
        uint8_t *ncand = (uint8_t *)ctx->ws2 + cand_bytes + rows * 64 * 8;
        // enough workgroups to fill the chip, each with enough rows to pay for staging the frame
        constexpr int WAVES = COLLECT_THREADS / 64;
        int wgs_per_inst = std::max(1, std::min((2048 + n_inst - 1) / n_inst, (waves_per_inst + 2 * WAVES - 1) / (2 * WAVES)));
        // few instances: one row of 64 queries per wave instead of two, so that an instance spreads over twice the workgroups
        const int groups = (long long)n_inst * wgs_per_inst < 256 ? 1 : 2;
        if (groups == 1) wgs_per_inst = std::max(1, std::min((2048 + n_inst - 1) / n_inst, (waves_per_inst + WAVES - 1) / WAVES));
        const int stage_nxt = collect_lds_bytes(max_n, true) <= 160 * 1024;
        const size_t lds = collect_lds_bytes(max_n, stage_nxt != 0);
        HIPCHK(ctx, (orbm_allow_dynamic_lds<&collect_kernel<MODE>>(ctx, lds)));
        hipLaunchKernelGGL((collect_kernel<MODE>), dim3((unsigned)(n_inst * wgs_per_inst)), dim3(COLLECT_THREADS), lds, ctx->stream, P,
                           waves_per_inst, wgs_per_inst, cand, cand_hi, ncand, postable, cap, stage_nxt, groups);
        HIPCHK(ctx, hipGetLastError());
        const int threads = max_q >= 1024 ? 1024 : (max_q > 256 ? 512 : 256);
        return launch_claims<MODE, false, true>(ctx, P, n_inst, max_n, threads, waves_per_inst, cand, ncand, postable, cap, cand_hi);
    }
    return ORBM_OK;
}

int launch_window(orbm_ctx *ctx, int mode, const WinParams &P, int n_inst, int max_n, int max_q) {
    switch (mode) {
        case MODE_LAST: return launch_mode<MODE_LAST>(ctx, P, n_inst, max_n, max_q);
        case MODE_KF: return launch_mode<MODE_KF>(ctx, P, n_inst, max_n, max_q);
        case MODE_SIM3: return launch_mode<MODE_SIM3>(ctx, P, n_inst, max_n, max_q);
        case MODE_INIT: return launch_mode<MODE_INIT>(ctx, P, n_inst, max_n, max_q);
        default: return launch_mode<MODE_PROJ>(ctx, P, n_inst, max_n, max_q);
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
    int max_n = 0, max_q = 0;
    if (!io_on_device) {
        int rc = orbm_check_frames(ctx, n_inst, frame, "frame");
        if (!rc) rc = orbm_check_qsets(ctx, n_inst, qset, mode == MODE_PROJ ? 0 : 1, "qset");
        if (rc) return rc;
        for (int k = 0; k < n_inst; k++) {
            max_n = std::max(max_n, ctx->h_frames[frame[k]].n);
            max_q = std::max(max_q, ctx->h_qsets[qset[k]].n);
        }
        if (max_n > stride) return orbm_fail(ctx, ORBM_EINVAL, "%s: stride %d < largest frame %d", who, stride, max_n);
    } else {
        max_n = stride;  // the caller's stride bounds every frame of the batch
        // the ids live on the device: bound the query count by the largest live query set of the context
        for (size_t k = 0; k < ctx->h_qsets.size(); k++)
            if (ctx->qset_arena[k] >= 0) max_q = std::max(max_q, ctx->h_qsets[k].n);
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
    rc = launch_window(ctx, mode, P, n_inst, std::max(max_n, 1), max_q);
    if (rc) return rc;
    return io.finish();
}

}  // namespace

// tuning / test knob (not part of the public ABI): see g_win_variant
extern "C" void orbm__window_variant(int v) { g_win_variant = v; }

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
    rc = launch_window(ctx, MODE_INIT, P, 1, std::max(n2, 1), n1);
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
