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
// (A query whose ratio test passes on distances now names and waits for its best slot only: the second-best can only get
// worse, see `second_matters`.)
//
// Three forms of that, chosen by launch_mode() from the shape of the call, all returning the same bits:
//   fused_kernel      one launch, one workgroup per instance, for batches (>= 48 instances) and for a5.  The search tests the
//                     slot BEFORE the descriptor, as the reference's own loop does, through free bitmaps over a per-launch LDS
//                     index; claims of a chunk are settled by all waves together or by wave 0 while the next chunk is searched.
//                     a5 searches cooperatively (a query's candidates over a wave's lanes).
//   collect_kernel +  the split form, for a call with one or a few instances: the search spread over the whole chip (every
//   window_kernel<PRE>  workgroup builds the LDS index, then one query per lane -- or, with few queries in the call, one or two
//                     queries per WAVE and a query's candidates over the lanes), then one workgroup per instance settles the claims
//                     on the candidate words the search left in global scratch.
//   window_kernel     the single-kernel forms (walk + rounds in one workgroup, frame optionally in LDS): frames whose index does
//                     not fit LDS, and the pass behind fused_kernel that redoes an instance one of whose queries overflowed.
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <type_traits>
#include <vector>
#include "orbm_internal.h"

namespace {

#ifdef ORBM_DIAG
__device__ unsigned long long d_win_stamps[128];   // timing only: wall-clock stamps of instance 0 of the split form's two kernels
#define WIN_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) d_win_stamps[(k)] = wall_clock64(); } while (0)
#else
#define WIN_STAMP(k) do { } while (0)
#endif

enum { MODE_LAST = 0, MODE_KF = 1, MODE_SIM3 = 2, MODE_INIT = 3, MODE_PROJ = 10 };

constexpr int WIN_MAX_N = 8160;       // slot state lives in LDS: 5 x 4 B per keypoint (+ 256 B) within 160 KB
constexpr uint32_t NOPOST = 0xFFFFFFFFu;

struct WinParams {
    const FrameDev *frames;
    const QSetDev *qsets;
    const int32_t *frame_ids;
    const int32_t *qset_ids;     // MODE_INIT: id of frame 1 (the query frame)
    const uint8_t *slot_flags;   // nullable, n_inst x stride
    const float *prev_xy;        // MODE_INIT only: vbPrevMatched as it comes in
    float *prev_out;             // ... and where the matched entries are written (:684-686); the others are left alone
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
// calls body(k, octave, uright, j) for every keypoint the reference would push, in the reference's order
// (ix ascending, iy ascending, insertion order); j = its position in the cell-ordered stream.  Cells (ix, minY..maxY) are consecutive in the CSR, so a
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
            const int oct = (int)((w >> 16) & 15u);
            if (bCheckLevels) {
                if (oct < minLevel) continue;
                if (maxLevel >= 0 && oct > maxLevel) continue;
            }
            const float distx = __fsub_rn(rc.x, x), disty = __fsub_rn(rc.y, y);
            if (fabsf(distx) < r && fabsf(disty) < r) body((int)(w & 0xFFFFu), oct, rc.z, (int)j);
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
    } else if (MODE == MODE_PROJ) {
        // one 16-byte record per query (QSetDev::pk): x, y, xr, level | flags << 8 | (viewCos >= 0.998f) << 16.  :168-174
        // compares the float viewCos with the double 0.998: true exactly for floats >= 0.998f (0x3F7F7CEE, whose value lies
        // above 0.998; the float below it lies below); the upload evaluates it
        const v4u k = ((gptr<const v4u>)S.QS.pk)[qi];
        q.flags = (k.w >> 8) & 0xFFu;
        if (q.flags & ORBM_Q_VALID) {
            q.valid = true;
            q.x = __uint_as_float(k.x); q.y = __uint_as_float(k.y); q.xr = __uint_as_float(k.z);
            const int lvl = (int)(k.w & 0xFFu);
            float rv = ((k.w >> 16) & 1u) ? 2.5f : 4.0f;
            if (P.th != 1.0f) rv = __fmul_rn(rv, P.th);                              // :61, :84-85
            q.r = __fmul_rn(rv, F.sf[lvl]);                                          // :89
            q.minL = lvl - 1; q.maxL = lvl;
        }
    } else {
        const QSetG &QS = S.QS;
        q.flags = QS.flags[qi];
        if (q.flags & ORBM_Q_VALID) {
            q.valid = true;
            q.x = QS.f0[qi]; q.y = QS.f1[qi];
            q.r = QS.f2[qi]; q.minL = QS.l0[qi]; q.maxL = QS.l1[qi];
            if (QS.f3) q.xr = QS.f3[qi];
            if (QS.angle) q.angle = QS.angle[qi];
        }
    }
    return q;
}

// window half-size of query qi (the same arithmetic as load_params), < 0 for a query that does not search
template <int MODE>
__device__ __forceinline__ float load_radius(const WinParams &P, const FrameG &F, const QSrc &S, int qi) {
    if (qi >= S.Q) return -1.f;
    if (MODE == MODE_INIT) return S.F1.octave[qi] == 0 ? (float)P.window : -1.f;
    if (MODE == MODE_PROJ) {
        const uint32_t w = S.QS.pk[4 * (size_t)qi + 3];
        if (!((w >> 8) & ORBM_Q_VALID)) return -1.f;
        float rv = ((w >> 16) & 1u) ? 2.5f : 4.0f;
        if (P.th != 1.0f) rv = __fmul_rn(rv, P.th);
        return __fmul_rn(rv, F.sf[w & 0xFFu]);
    }
    if (!(S.QS.flags[qi] & ORBM_Q_VALID)) return -1.f;
    return S.QS.f2[qi];
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

// ---- split form, kernel 1 of 2: ONE window search per query, anywhere on the chip.  Nothing here depends on the slot
// state, so queries are independent.  A workgroup first builds, in LDS, an index of the frame made for THIS launch's
// windows: one structure per octave pair {p-1, p} (a query's level range is one such pair in a6, a5 and the Sim3 overload;
// the overloads with wider ranges get ONE structure holding every octave), each with coarse columns of 2^k fine grid
// columns (k chosen on the host from the launch's window radius: a window then spans at most ~4 coarse columns) and the 48
// fine rows.  Records are bucketed by (coarse column, fine row): rows y0..y1 of one coarse column are ONE contiguous run of
// 16-byte records, every one of which passes the level gate.  A query (one per lane) walks its few runs in a single flat
// loop (a column change is just another iteration), parks the survivors of the exact gates of Frame::GetFeaturesInArea
// (cell column inside [x0, x1], |dx| < r, |dy| < r; Frame.cc:434-498) and the stereo gate, then a second, wave-uniform
// loop computes their Hamming distances.  The walk visits records in bucket order, not in the reference's visit order
// (ix, iy, index in cell): the order is carried by the keys instead.  A candidate is the word
//     dist << 17 | j << 4 | octave        j = the keypoint's position in the reference's cell-ordered stream (cell_idx)
// so ascending words = ascending (distance, reference visit order): exactly the order in which the reference's strict-'<'
// scan ranks its candidates, so the claims kernel can take them in any order.  Per query the FOUR smallest words go, sorted,
// into `head` (16 bytes, written for every query: the claims kernel's liveness scan reads nothing else); up to `tcap` more
// go, unsorted, into `tail` (4-byte stores, only for queries with more than four candidates).  Bits 28-31 of head[3] and
// head[2] = number of tail words (8 bits); 255 = more candidates than the tail holds (their distances are unknown: such a
// query counts as live and the claims kernel walks its window again).  Survivors are parked in LDS (the first PARK) and
// in the query's own tail slots (the rest) between the two phases.
constexpr int COLLECT_THREADS = 1024;
constexpr int COST_CLASSES = 64;
constexpr int PARK = 16;                    // survivors parked per query in LDS (u16 record positions); = WIN_CACHE
constexpr uint32_t W_INVALID = 0xFFFFFFFFu; // no candidate (bit 27 set; valid words stay below 2^27)
constexpr uint32_t W_FLAGS = 15u << 28;     // bits 28-31 of head[2] and head[3] carry the tail count
constexpr int TAIL_UNKNOWN = 255;           // tail count: more candidates than the tail holds

struct CoarseCfg {
    int nstruct;          // 8: one structure per octave pair {p-1, p}; 1: every octave in one structure
    int kshift[8];        // coarse column = fine column >> kshift
    int base[9];          // first bucket of structure s (bucket = base + coarse column * 48 + fine row); base[nstruct] = all buckets
    int rec_cap;          // records the index may hold (2n with pairs, n otherwise)
    int n_cap;            // keypoints per frame the launch was sized for
};

__device__ __forceinline__ uint32_t word_slot(uint32_t v) { return (v >> 4) & 0x1FFFu; }
__device__ __forceinline__ int word_oct(uint32_t v) { return (int)(v & 15u); }
__device__ __forceinline__ int word_dist(uint32_t v) { return (int)((v & ~W_FLAGS) >> 17); }
__device__ __forceinline__ bool word_valid(uint32_t v) { return !(v & (1u << 27)); }
__device__ __forceinline__ int word_tail(uint32_t head2, uint32_t head3) {
    return word_valid(head3) ? (int)((head3 >> 28) | ((head2 >> 28) << 4)) : 0;
}

template <int MODE>
__global__ __launch_bounds__(COLLECT_THREADS) void collect_kernel(WinParams P, int waves_per_inst, int wgs_per_inst,
                                                                  v4u *head_out, uint32_t *tail_out, int tcap, CoarseCfg cfg, int groups, int qt, int cost_exact, int desc_lds,
                                                                  int coop) {
    extern __shared__ __align__(16) uint32_t lds[];
    __shared__ int s_k[8], s_base[9];
    __shared__ uint32_t s_part[COLLECT_THREADS / 64];
    const int inst = blockIdx.x / wgs_per_inst, part = blockIdx.x - inst * wgs_per_inst;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const FrameG F(P.frames[P.frame_ids[inst]]);
    const int total = cfg.base[cfg.nstruct];                                     // buckets
    const int offn = (total + 6) & ~1;                                           // u16 entries of c_off (see below), even
    uint16_t *c_off = reinterpret_cast<uint16_t *>(lds);
    v4f *c_rec = reinterpret_cast<v4f *>(lds + ((offn / 2 + 3) & ~3));           // one record per stream position
    uint16_t *c_lst = reinterpret_cast<uint16_t *>(c_rec + ((cfg.n_cap + 3) & ~3));   // the buckets' runs: stream positions
    uint16_t *l_cc = c_lst + ((cfg.rec_cap + 7) & ~7);                           // PARK x COLLECT_THREADS stream positions
    int *l_bucket = (int *)(l_cc + PARK * COLLECT_THREADS);                      // cost classes
    uint16_t *l_perm = reinterpret_cast<uint16_t *>(l_bucket + COST_CLASSES);    // 2 x COLLECT_THREADS
    v4u *l_desc = reinterpret_cast<v4u *>(l_perm + 2 * COLLECT_THREADS + 32);    // desc_lds: the frame's descriptors in stream order
    if (tid < 8) s_k[tid] = cfg.kshift[tid];
    if (tid < 9) s_base[tid] = cfg.base[tid];
    // ---- the index.  Bucket b's run is c_rec[c_off[b + 1] .. c_off[b + 2]): counters sit at c_off[b + 2] (two u16 per
    // atomic word), are turned into start positions by an exclusive scan and advanced to END positions by the scatter.
    uint32_t *cnt32 = lds;
    WIN_STAMP(0);
    for (int i = tid; i < offn / 2; i += COLLECT_THREADS) cnt32[i] = 0;
    __syncthreads();
    const int T = (int)F.cell_off[ORBM_GRID_CELLS];
    const bool pairs = cfg.nstruct > 1;
    auto bucket_add = [&](int b) {                                               // -> value of bucket b's u16 before the add
        const uint32_t old = atomicAdd(&cnt32[(b + 2) >> 1], ((b + 2) & 1) ? 0x10000u : 1u);
        return ((b + 2) & 1) ? old >> 16 : old & 0xFFFFu;
    };
    for (int j = tid; j < T; j += COLLECT_THREADS) {
        const uint32_t w = __float_as_uint(F.cell_rec[j].w);
        const int oct = (int)((w >> 16) & 15u), cy = (int)((w >> 20) & 63u), cx = (int)(w >> 26);
        if (pairs) {
            if (oct <= 7) bucket_add(s_base[oct] + (cx >> s_k[oct]) * ORBM_GRID_ROWS + cy);
            if (oct + 1 <= 7) bucket_add(s_base[oct + 1] + (cx >> s_k[oct + 1]) * ORBM_GRID_ROWS + cy);
        } else {
            bucket_add((cx >> s_k[0]) * ORBM_GRID_ROWS + cy);
        }
    }
    __syncthreads();
    WIN_STAMP(1);
    {
        const int per = (total + COLLECT_THREADS - 1) / COLLECT_THREADS;
        const int i0 = min(total, tid * per), i1 = min(total, i0 + per);
        uint32_t sum = 0;
        for (int i = i0; i < i1; i++) sum += c_off[i + 2];
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (lane == 63) s_part[wave] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (int w2 = 0; w2 < wave; w2++) run += s_part[w2];
        for (int i = i0; i < i1; i++) { const uint32_t c = c_off[i + 2]; c_off[i + 2] = (uint16_t)run; run += c; }
        if (tid == 0) { c_off[0] = 0; c_off[1] = 0; }
    }
    __syncthreads();
    WIN_STAMP(2);
    for (int j = tid; j < T; j += COLLECT_THREADS) {
        const v4f rc = F.cell_rec[j];
        const uint32_t w = __float_as_uint(rc.w);
        const int oct = (int)((w >> 16) & 15u), cy = (int)((w >> 20) & 63u), cx = (int)(w >> 26);
        c_rec[j] = v4f{rc.x, rc.y, rc.z, __uint_as_float((uint32_t)j | ((uint32_t)oct << 13) | ((uint32_t)cx << 17))};
        if (pairs) {
            if (oct <= 7) c_lst[bucket_add(s_base[oct] + (cx >> s_k[oct]) * ORBM_GRID_ROWS + cy)] = (uint16_t)j;
            if (oct + 1 <= 7) c_lst[bucket_add(s_base[oct + 1] + (cx >> s_k[oct + 1]) * ORBM_GRID_ROWS + cy)] = (uint16_t)j;
        } else {
            c_lst[bucket_add((cx >> s_k[0]) * ORBM_GRID_ROWS + cy)] = (uint16_t)j;
        }
        if (desc_lds) {                      // the candidates' descriptors are read from LDS, by stream position
            const int kp = (int)(w & 0xFFFFu);
            const gptr<const v4u> gd = (gptr<const v4u>)F.desc;
            l_desc[2 * j] = gd[2 * kp]; l_desc[2 * j + 1] = gd[2 * kp + 1];
        }
    }
    __syncthreads();
    WIN_STAMP(3);
    const QSrc S = query_source<MODE>(P, inst);
    const bool stereo_gate = has_stereo_gate<MODE>(S);
    const gptr<const v4u> gdesc = (gptr<const v4u>)F.desc;
    const float min_x = F.min_x, min_y = F.min_y, inv_w = F.inv_w, inv_h = F.inv_h;
    if (coop) {
        // ---- few queries in the whole launch (one tracking call): `coop` queries per workgroup, one or two per WAVE, and a
        // wave spreads a query's candidates over its lanes -- the runs of the window's coarse columns laid end to end (a
        // prefix sum over the columns, one per lane), lane l takes entry l: gates, distance, word.  One query per lane made
        // a wave walk its longest window, fifteen to twenty-five dependent steps, while the other fifteen waves of the
        // workgroup had nothing to do (13 of the kernel's 19 us for a 1500-query call).  The four smallest words are picked
        // with four wave minima; more than 64 survivors (or a level range the index does not cover): the query is
        // flagged and the claims kernel walks its window itself, as for any query with more candidates than the tail holds.
        const int per_wave = coop / (COLLECT_THREADS / 64), nq_inst = waves_per_inst * 64;
        uint32_t *wscr = reinterpret_cast<uint32_t *>(l_cc) + wave * 64;
        Query qs[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int qi = part * coop + wave * per_wave + u;
            qs[u] = load_params<MODE>(P, F, S, (u < per_wave && qi < nq_inst) ? qi : S.Q);
            if (qs[u].valid) load_desc<MODE>(S, qi, qs[u]);
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int qi = part * coop + wave * per_wave + u;
            if (u >= per_wave || qi >= nq_inst) continue;
            const Query &q = qs[u];
            uint32_t k0 = W_INVALID, k1 = W_INVALID, k2 = W_INVALID, k3 = W_INVALID;
            bool rewalk = false;
            int nsel = 0;
            uint32_t *mytail = tail_out + ((size_t)inst * ((size_t)waves_per_inst * 64) + qi) * tcap;
            const float xr = __fsub_rn(q.x, min_x), yr = __fsub_rn(q.y, min_y), r = q.r;
            const int x0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(xr, r), inv_w)));
            const int x1 = min(ORBM_GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(xr, r), inv_w)));
            const int y0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(yr, r), inv_h)));
            const int y1 = min(ORBM_GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(yr, r), inv_h)));
            int lo = -1, hi = 255;
            if (MODE == MODE_SIM3) { lo = q.minL; hi = q.maxL; }
            else if (q.minL > 0 || q.maxL >= 0) { lo = q.minL; if (q.maxL >= 0) hi = q.maxL; }
            const bool open_win = q.valid && x0 < ORBM_GRID_COLS && x1 >= 0 && y0 < ORBM_GRID_ROWS && y1 >= 0 && x1 >= x0 && y1 >= y0;
            if (__builtin_amdgcn_readfirstlane((int)open_win)) {
                int st = 0;
                bool covered = !pairs;
                if (pairs && hi <= 7 && hi >= 0 && hi - lo <= 1) { st = hi; covered = true; }
                st = __builtin_amdgcn_readfirstlane(st);
                if (!__builtin_amdgcn_readfirstlane((int)covered)) rewalk = true;
                else {
                    const uint32_t qd[8] = {q.d0.x, q.d0.y, q.d0.z, q.d0.w, q.d1.x, q.d1.y, q.d1.z, q.d1.w};
                    const int kk = s_k[st], sb = s_base[st];
                    const int X0 = __builtin_amdgcn_readfirstlane(x0 >> kk), ncol = __builtin_amdgcn_readfirstlane((x1 >> kk) - (x0 >> kk) + 1);
                    uint32_t jc = 0, len = 0;
                    if (lane < ncol) {
                        const int b = sb + (X0 + lane) * ORBM_GRID_ROWS;
                        jc = c_off[b + y0 + 1];
                        len = (uint32_t)c_off[b + y1 + 2] - jc;
                    }
                    uint32_t incl = len;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        const uint32_t up = __shfl_up(incl, d, 64);
                        if (lane >= d) incl += up;
                    }
                    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    auto candidate = [&](uint32_t l) -> uint32_t {
                        if (l >= total) return W_INVALID;
                        uint32_t idx = 0;
                        for (int c2 = 0; c2 < ncol; c2++) {
                            const uint32_t ic = (uint32_t)__builtin_amdgcn_readlane((int)incl, c2), lc = (uint32_t)__builtin_amdgcn_readlane((int)len, c2);
                            const uint32_t j0 = (uint32_t)__builtin_amdgcn_readlane((int)jc, c2);
                            if (l < ic && l >= ic - lc) idx = j0 + (l - (ic - lc));
                        }
                        const uint32_t jcur = c_lst[idx];
                        const v4f rc = c_rec[jcur];
                        const uint32_t w = __float_as_uint(rc.w);
                        const int oct = (int)((w >> 13) & 15u), cx = (int)(w >> 17);
                        int pass = (int)(cx >= x0) & (int)(cx <= x1) & (int)(oct >= lo) & (int)(oct <= hi) &
                                   (int)(fabsf(__fsub_rn(rc.x, q.x)) < r) & (int)(fabsf(__fsub_rn(rc.y, q.y)) < r);
                        if (stereo_gate) pass &= 1 ^ ((int)(rc.z > 0) & (int)(fabsf(__fsub_rn(q.xr, rc.z)) > r));   // :117-122 / :1883-1891
                        if (!pass) return W_INVALID;
                        v4u dlo, dhi;
                        if (desc_lds) { dlo = l_desc[2 * jcur]; dhi = l_desc[2 * jcur + 1]; }
                        else { const int kp = (int)F.cell_idx[jcur]; dlo = gdesc[2 * kp]; dhi = gdesc[2 * kp + 1]; }
                        const uint32_t cd[8] = {dlo.x, dlo.y, dlo.z, dlo.w, dhi.x, dhi.y, dhi.z, dhi.w};
                        return (hamming256(qd, cd) << 17) | (jcur << 4) | (uint32_t)oct;
                    };
                    uint32_t word = W_INVALID;
                    if (total <= 64) word = candidate((uint32_t)lane);
                    else {
                        // (wide windows over crowded cells: the survivors of every pass are gathered in LDS first)
                        int nsurv = 0;
                        for (uint32_t eb = 0; eb < total; eb += 64) {
                            const uint32_t w2 = candidate(eb + (uint32_t)lane);
                            const unsigned long long sm = __ballot(w2 != W_INVALID);
                            if (w2 != W_INVALID) {
                                const int pp = nsurv + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(sm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sm, 0u));
                                if (pp < 64) wscr[pp] = w2;
                            }
                            nsurv += (int)__popcll(sm);
                        }
                        if (nsurv > 64) rewalk = true;
                        else {
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            if (lane < nsurv) word = wscr[lane];
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                    if (!rewalk) {
                        nsel = (int)__popcll(__ballot(word != W_INVALID));
                        uint32_t rest = word, kk4[4];
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            uint32_t m = rest;
#pragma unroll
                            for (int off = 32; off > 0; off >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, off, 64));
                            kk4[i] = m;
                            if (rest == m) rest = W_INVALID;          // (words are distinct: they carry the stream position)
                        }
                        k0 = kk4[0]; k1 = kk4[1]; k2 = kk4[2]; k3 = kk4[3];
                        const unsigned long long rm = __ballot(rest != W_INVALID);
                        if (rest != W_INVALID) {
                            const int pp = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(rm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)rm, 0u));
                            if (pp < tcap) mytail[pp] = rest;
                        }
                        if (nsel > tcap + 4) rewalk = true;
                    }
                }
            }
            if (lane == 0) {
                if (nsel > 4) {
                    const uint32_t tc = (uint32_t)(nsel - 4);
                    k3 |= (tc & 15u) << 28; k2 |= (tc >> 4) << 28;
                }
                if (rewalk) { k2 = W_FLAGS; k3 = W_FLAGS; }              // count 255: live, re-walked by the claims kernel
                head_out[(size_t)inst * ((size_t)waves_per_inst * 64) + qi] = v4u{k0, k1, k2, k3};
            }
        }
        WIN_STAMP(4); WIN_STAMP(5);
        return;
    }
    // queries sorted together: `groups` rows of 64 per wave (2; 1 when the launch has few instances, so that one instance
    // spreads over more workgroups)
    // qt = queries per row of a super-chunk (a multiple of 64, COLLECT_THREADS unless the launch has too few queries to give
    // every CU a full workgroup's worth: then only the first qt / 64 waves take queries and the instance spreads wider)
    const int SUPER = groups * qt, QW = qt >> 6;
    // A wave costs what its longest walk costs and window areas differ by more than 10x between pyramid levels, so
    // the queries of a super-chunk are counting-sorted by window area (in cells) and each wave takes one heavy and one
    // light group of 64 (results are written at the query's own position).
    const int n_super = (waves_per_inst * 64 + SUPER - 1) / SUPER;
    const size_t qstride = (size_t)waves_per_inst * 64;
    for (int sc = part; sc < n_super; sc += wgs_per_inst) {
        if (tid < COST_CLASSES) l_bucket[tid] = 0;
        __syncthreads();
        int cls[2], rank[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (h >= groups || tid >= qt) { cls[h] = 0; rank[h] = 0; continue; }
            // cost class = the trip count of the query's walk below: the records of its runs plus one per coarse column
            // (the window arithmetic and the run bounds are computed here as well as there; exact classes keep the lanes
            // of a wave together, a class by window area left 46 % of them idle)
            int c = 0;
            if (cost_exact & 1) {
                const Query q = load_params<MODE>(P, F, S, sc * SUPER + h * qt + tid);
                if (q.valid) {
                    const float xr = __fsub_rn(q.x, min_x), yr = __fsub_rn(q.y, min_y), r = q.r;
                    const int x0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(xr, r), inv_w)));
                    const int x1 = min(ORBM_GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(xr, r), inv_w)));
                    const int y0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(yr, r), inv_h)));
                    const int y1 = min(ORBM_GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(yr, r), inv_h)));
                    int lo = -1, hi = 255;
                    if (MODE == MODE_SIM3) { lo = q.minL; hi = q.maxL; }
                    else if (q.minL > 0 || q.maxL >= 0) { lo = q.minL; if (q.maxL >= 0) hi = q.maxL; }
                    int st = 0;
                    bool covered = !pairs;
                    if (pairs && hi <= 7 && hi >= 0 && hi - lo <= 1) { st = hi; covered = true; }
                    if (covered && x1 >= x0 && y1 >= y0) {
                        const int k = s_k[st], sb = s_base[st];
                        int trip = 1;
                        for (int X = x0 >> k; X <= (x1 >> k); X++) {
                            const int b = sb + X * ORBM_GRID_ROWS;
                            trip += 1 + (int)c_off[b + y1 + 2] - (int)c_off[b + y0 + 1];
                        }
                        c = trip < 48 ? trip : min(COST_CLASSES - 1, 48 + ((trip - 48) >> 4));
                    }
                }
            } else {
                const float rr = load_radius<MODE>(P, F, S, sc * SUPER + h * qt + tid);
                if (rr >= 0.f) {
                    const int area = (int)((2.f * rr * inv_w + 2.f) * (2.f * rr * inv_h + 2.f));
                    c = area < 124 ? 1 + (max(area, 0) >> 1) : min(COST_CLASSES - 1, 62 + ((area - 124) >> 6));
                }
            }
            cls[h] = c;
            rank[h] = atomicAdd(&l_bucket[c], 1);
        }
        __syncthreads();
        if (wave == 0) {                                          // class c starts after every heavier class: suffix sums, once
            const int mine = l_bucket[lane];
            int incl = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int dn = __shfl_down(incl, d, 64);
                if (lane + d < 64) incl += dn;
            }
            l_bucket[lane] = incl - mine;
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (h >= groups || tid >= qt) continue;
            l_perm[l_bucket[cls[h]] + rank[h]] = (uint16_t)(h * qt + tid);
        }
        __syncthreads();
        WIN_STAMP(4);
      for (int h = 0; h < groups && wave < QW && !(cost_exact & 8); h++) {
        const int group = h == 0 ? wave : 2 * QW - 1 - wave;
        const int qi = sc * SUPER + (int)l_perm[group * 64 + lane];
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
        int ncand = 0;      // candidates that passed every gate
        bool uncovered = false;
        uint32_t *mytail = tail_out + ((size_t)inst * qstride + min(qi, (int)qstride - 1)) * tcap;
        // first phase: a single flat loop over the runs of the window's coarse columns
        if (!(cost_exact & 2) && q.valid && x0 < ORBM_GRID_COLS && x1 >= 0 && y0 < ORBM_GRID_ROWS && y1 >= 0 && x1 >= x0 && y1 >= y0) {   // (bits 1-3 of cost_exact: timing diagnostics)
            // the structure of the octave pair {hi-1, hi} holds every octave of [lo, hi] when hi - lo <= 1; a wider (or
            // open) range can only come with the one-structure index
            int st = 0;
            bool covered = !pairs;
            if (pairs && hi <= 7 && hi >= 0 && hi - lo <= 1) { st = hi; covered = true; }
            const int k = s_k[st], sb = s_base[st];
            int X = (x0 >> k) - 1;
            const int X1 = x1 >> k;
            uint32_t j = 0, e = 0;
            bool more = covered;
            uncovered = !covered;                                // cannot happen with the launch rules below; if it does: re-walk
            while (more) {
                if (j < e) {
                    const uint32_t jcur = c_lst[j];               // stream position of the run's next record
                    const v4f rc = c_rec[jcur];
                    j++;
                    const uint32_t w = __float_as_uint(rc.w);
                    const int oct = (int)((w >> 13) & 15u), cx = (int)(w >> 17);
                    int pass = (int)(cx >= x0) & (int)(cx <= x1) & (int)(oct >= lo) & (int)(oct <= hi) &
                               (int)(fabsf(__fsub_rn(rc.x, q.x)) < r) & (int)(fabsf(__fsub_rn(rc.y, q.y)) < r);
                    if (stereo_gate) pass &= 1 ^ ((int)(rc.z > 0) & (int)(fabsf(__fsub_rn(q.xr, rc.z)) > r));   // :117-122 / :1883-1891
                    if (pass) {
                        if (ncand < PARK) l_cc[ncand * COLLECT_THREADS + tid] = (uint16_t)jcur;
                        else if (ncand < tcap + 4) mytail[ncand - 4] = jcur;   // parked where its word will go
                        ncand++;
                    }
                }
                if (j >= e) {                                  // run finished (or empty): the next coarse column, in the same iteration
                    if (++X > X1) more = false;
                    else {
                        const int b = sb + X * ORBM_GRID_ROWS;
                        j = c_off[b + y0 + 1];
                        e = c_off[b + y1 + 2];
                    }
                }
                // keeps this ONE loop: without a convergent operation here the compiler splits it into nested
                // "advance until a survivor" loops, and lanes then wait for each other at every survivor
                __builtin_amdgcn_wave_barrier();
            }
        }
        // second phase: Hamming distances, wave-uniform trip count; the four smallest words sorted, the others to the tail
        uint32_t k0 = W_INVALID, k1 = W_INVALID, k2 = W_INVALID, k3 = W_INVALID;
        {
            const int nhave = (cost_exact & 4) ? 0 : min(ncand, tcap + 4);
            int wn = nhave;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) wn = max(wn, __shfl_xor(wn, off, 64));
            wn = __builtin_amdgcn_readfirstlane(wn);
            for (int i = 0; i < wn; i++)
                if (i < nhave) {
                    const uint32_t pos = i < PARK ? (uint32_t)l_cc[i * COLLECT_THREADS + tid] : mytail[i - 4];
                    const uint32_t w = __float_as_uint(c_rec[pos].w);
                    const uint32_t j13 = pos;
                    v4u dlo, dhi;
                    if (desc_lds) { dlo = l_desc[2 * pos]; dhi = l_desc[2 * pos + 1]; }
                    else { const int kp = (int)F.cell_idx[pos]; dlo = gdesc[2 * kp]; dhi = gdesc[2 * kp + 1]; }
                    const uint32_t cd[8] = {dlo.x, dlo.y, dlo.z, dlo.w, dhi.x, dhi.y, dhi.z, dhi.w};
                    uint32_t x = (hamming256(qd, cd) << 17) | (j13 << 4) | ((w >> 13) & 15u), t;
                    t = min(k0, x); x = max(k0, x); k0 = t;
                    t = min(k1, x); x = max(k1, x); k1 = t;
                    t = min(k2, x); x = max(k2, x); k2 = t;
                    t = min(k3, x); x = max(k3, x); k3 = t;
                    if (i >= 4) mytail[i - 4] = x;               // what fell out of the four (a real word from the fifth candidate on)
                }
        }
        if (qi < waves_per_inst * 64) {
            if (ncand > 4) {
                const uint32_t tc = ncand > tcap + 4 ? (uint32_t)TAIL_UNKNOWN : (uint32_t)(ncand - 4);
                k3 |= (tc & 15u) << 28; k2 |= (tc >> 4) << 28;
            }
            if (uncovered) { k2 = W_FLAGS; k3 = W_FLAGS; }       // no candidate words, count 255: live, re-walked by the claims kernel
            head_out[(size_t)inst * qstride + qi] = v4u{k0, k1, k2, k3};
        }
      }
        __syncthreads();     // l_perm and l_bucket are rewritten by the next super-chunk
        WIN_STAMP(5);
    }
}

// ---- fused form: ONE launch, one 1024-thread workgroup per (frame, query set) instance, queries taken in index order
// FUSED_CHUNK at a time.  Waves 1..15 search (one query per lane), wave 0 settles the claims of the chunk before while
// they do.  What makes this the fast form for the searches that write slot state (a6, a11, a5) is what the reference's own
// loop does first: it tests the slot (`if(F.mvpMapPoints[idx]) if(...Observations()>0) continue;`, ORBmatcher.cc:113-115,
// :1879-1881, :2048, :512) BEFORE it looks at the descriptor.  Slots only ever become blocked, and the state a searching
// wave reads holds claims of LOWER-index queries only (claims of chunk c are applied while chunk c + 1 is searched, claims
// of chunk c + 1 start after its search has ended), so a record that reads "blocked" is blocked for this query in the
// sequential order too and is dropped before the float gates, the descriptor gather and the distance; one that reads
// "free" is looked at again by the rounds.  Once the frame's slots have filled up -- 50 000 map points into 2000
// keypoints: after the first few thousand queries -- a query costs its walk over the index and nothing else: no
// descriptor load (its own 32 bytes are requested only when a free record passes the gates), no distance, no record
// anywhere.  A query with a free candidate within the threshold is LIVE: its four smallest candidate words, sorted, go to
// a pending list in LDS (further words to a per-thread tail in global scratch, rarely), and wave 0 runs the same
// evaluate / contested / decide rounds as window_kernel on up to 64 live queries at a time, one per lane, in index order,
// wave-synchronously: no workgroup barrier inside the rounds, and a post is (round stamp << 6 | 63 - lane) under
// atomicMax, so the post array never needs clearing.  Two workgroup barriers per chunk hand the pending list over.
constexpr int FUSED_THREADS = 1024;
constexpr int FUSED_CHUNK = FUSED_THREADS - 64;   // queries per chunk: waves 1..15
constexpr int PEND_WORDS = 10;                    // pending entry: meta, the query's angle, eight candidate words
constexpr int FPARK = 2;                          // survivors a lane parks (in registers) between two distance passes

// a5 (fused_kernel<MODE_INIT>) keeps, beyond the index: per query of a chunk its position, window and result (16 B) and its
// descriptor (32 B), and the frame's descriptors by stream position (32 B per keypoint) -- see the cooperative search there
static size_t fused_init_extra_bytes(const CoarseCfg &c) { return (size_t)(FUSED_THREADS - 64) * 48 + (size_t)c.n_cap * 32; }
static size_t fused_lds_bytes(const CoarseCfg &c, bool with_binmask) {
    const int total = c.base[c.nstruct];
    const size_t offn = (size_t)((total + 6) & ~1);
    const size_t n4 = (size_t)((c.n_cap + 3) & ~3);
    return (((offn / 2 + 3) & ~(size_t)3)) * 4 + n4 * 16 + (size_t)((c.rec_cap + 7) & ~7) * 2 + (size_t)((c.n_cap + 7) & ~7) * 2 +
           n4 * 4 * (3 + (with_binmask ? 1 : 0)) + (size_t)64 * PEND_WORDS * 4 +
           (size_t)(((c.rec_cap + 31) / 32 + 4) & ~3) * 4 /* fbits */ + (8 * 12 + 8 * 96) * 4 /* fblk, fcnt */ + 64;
}

// raw per-query parameters as loaded (requested a chunk ahead), decoded when the chunk is searched
struct RawQ {
    v4u pk;                      // MODE_PROJ
    float f0, f1, f2, f3, ang;   // other modes: u, v, radius, ur, angle | MODE_INIT: prev x, prev y, -, -, angle
    int l0, l1;
    uint32_t flags;              // MODE_INIT: octave
    bool in_range;
};

template <int MODE>
__device__ __forceinline__ RawQ raw_load(const WinParams &P, const QSrc &S, int qi) {
    RawQ r;
    r.pk = v4u{0, 0, 0, 0}; r.f0 = r.f1 = r.f2 = r.f3 = r.ang = 0.f; r.l0 = r.l1 = -1; r.flags = 0;
    r.in_range = qi < S.Q;
    if (!r.in_range) return r;
    if (MODE == MODE_INIT) {
        r.flags = S.F1.octave[qi];
        r.f0 = P.prev_xy[2 * qi]; r.f1 = P.prev_xy[2 * qi + 1];
        r.ang = S.F1.angle[qi];
    } else if (MODE == MODE_PROJ) {
        r.pk = ((gptr<const v4u>)S.QS.pk)[qi];
    } else {
        const QSetG &QS = S.QS;
        r.flags = QS.flags[qi];
        r.f0 = QS.f0[qi]; r.f1 = QS.f1[qi]; r.f2 = QS.f2[qi]; r.l0 = QS.l0[qi]; r.l1 = QS.l1[qi];
        if (QS.f3) r.f3 = QS.f3[qi];
        if (QS.angle) r.ang = QS.angle[qi];
    }
    return r;
}

// = load_params() on the loaded values; sf = the frame's scale factors (in LDS)
template <int MODE>
__device__ __forceinline__ Query raw_decode(const WinParams &P, const RawQ &r, const float *sf) {
    Query q;
    q.d0 = q.d1 = v4u{0, 0, 0, 0};
    q.x = q.y = q.r = q.xr = q.angle = 0.f; q.minL = q.maxL = -1; q.ql = 0; q.flags = 0; q.valid = false;
    if (!r.in_range) return q;
    if (MODE == MODE_INIT) {
        if (r.flags == 0) {                                                           // :582-584
            q.valid = true;
            q.x = r.f0; q.y = r.f1;
            q.r = (float)P.window; q.minL = 0; q.maxL = 0;                            // :587
            q.angle = r.ang;
        }
    } else if (MODE == MODE_PROJ) {
        const v4u k = r.pk;
        q.flags = (k.w >> 8) & 0xFFu;
        if (q.flags & ORBM_Q_VALID) {
            q.valid = true;
            q.x = __uint_as_float(k.x); q.y = __uint_as_float(k.y); q.xr = __uint_as_float(k.z);
            const int lvl = (int)(k.w & 0xFFu);
            float rv = ((k.w >> 16) & 1u) ? 2.5f : 4.0f;
            if (P.th != 1.0f) rv = __fmul_rn(rv, P.th);                              // :61, :84-85
            q.r = __fmul_rn(rv, sf[lvl & (ORBM_MAX_LEVELS - 1)]);                     // :89
            q.minL = lvl - 1; q.maxL = lvl;
        }
    } else {
        q.flags = r.flags;
        if (q.flags & ORBM_Q_VALID) {
            q.valid = true;
            q.x = r.f0; q.y = r.f1; q.r = r.f2; q.minL = r.l0; q.maxL = r.l1; q.xr = r.f3; q.angle = r.ang;
        }
    }
    return q;
}

// workgroup barrier that orders LDS accesses only: the parameter loads a wave has requested for the NEXT chunk stay in flight
// across it (__syncthreads() would wait for them: its fence covers global memory too)
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int MODE>
__global__ __launch_bounds__(FUSED_THREADS) void fused_kernel(WinParams P, CoarseCfg cfg, uint32_t *tail_scratch, int tcap, int32_t *redo_flags,
                                                                unsigned long long *trace) {
    extern __shared__ __align__(16) uint32_t lds[];
    __shared__ int s_k[8], s_base[9];
    __shared__ uint32_t s_part[FUSED_THREADS / 64];
    __shared__ unsigned long long s_live[16];
    __shared__ int s_hist[36];                                                    // hist[32], nmatch, keep[3]
    __shared__ float s_sf[ORBM_MAX_LEVELS];
    __shared__ int s_any[3];
    __shared__ uint32_t s_stamp_mem;
    __shared__ int s_free_total;                                                  // free (not blocked) slots of the frame
    uint32_t *s_stamp = &s_stamp_mem;
    constexpr int NARROW_MAX = 64;                                                // up to this many live queries of a chunk go to wave 0
    constexpr bool HIST_MODE = MODE == MODE_LAST || MODE == MODE_KF || MODE == MODE_INIT;
    constexpr bool USES_SECOND = MODE == MODE_PROJ || MODE == MODE_INIT;
    constexpr int NW = 8;                                                         // candidate words in a pending entry
    const int inst = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const FrameG F(P.frames[P.frame_ids[inst]]);
    const int n = F.n;
    gptr<int32_t> redo_flag = in_global(redo_flags) + inst;
    if (tid == 0) *redo_flag = 0;
    const int total = cfg.base[cfg.nstruct];
    const int offn = (total + 6) & ~1;
    const int n4 = (cfg.n_cap + 3) & ~3;
    const int fwords = ((cfg.rec_cap + 31) / 32 + 4) & ~3;
    uint16_t *c_off = reinterpret_cast<uint16_t *>(lds);
    v4f *c_rec = reinterpret_cast<v4f *>(lds + ((offn / 2 + 3) & ~3));
    uint16_t *c_lst = reinterpret_cast<uint16_t *>(c_rec + n4);
    uint16_t *l_kp = c_lst + ((cfg.rec_cap + 7) & ~7);                            // stream position -> keypoint
    uint32_t *state = reinterpret_cast<uint32_t *>(l_kp + ((cfg.n_cap + 7) & ~7));   // ORBM_MP_* flags | vMatchedDistance, by stream position
    int32_t *owner = (int32_t *)(state + n4);
    uint32_t *post = state + 2 * (size_t)n4;
    uint32_t *binmask = state + 3 * (size_t)n4;                                   // HIST_MODE only
    uint32_t *pending = state + (3 + (HIST_MODE ? 1 : 0)) * (size_t)n4;
    uint32_t *fbits = pending + (size_t)64 * PEND_WORDS;                          // bit p: the record at c_lst[p] is free (not blocked)
    // fblk[s * 12 + BY] bit BX: structure s holds a free record in the block of 4 x 4 grid cells (BX, BY); fcnt = how many
    // (two 16-bit counters per word).  A query whose window touches no such block is finished without a walk.
    uint32_t *fblk = fbits + fwords;
    uint32_t *fcnt = fblk + 8 * 12;
    // a5 only (see the cooperative search in the chunk loop)
    float *s_qx = reinterpret_cast<float *>(fcnt + 8 * 96), *s_qy = s_qx + FUSED_CHUNK;
    uint32_t *s_qwin = reinterpret_cast<uint32_t *>(s_qy + FUSED_CHUNK), *s_qres = s_qwin + FUSED_CHUNK;
    v4u *s_qdesc = reinterpret_cast<v4u *>(s_qres + FUSED_CHUNK);
    v4u *ldesc = s_qdesc + 2 * FUSED_CHUNK;
    int *hist = s_hist, *nmatch = s_hist + 32, *keep = s_hist + 33;
#ifdef ORBM_DIAG
    if (trace && inst == 0 && tid == 0) trace[0] = wall_clock64();
#endif
    if (tid < 8) s_k[tid] = cfg.kshift[tid];
    if (tid < 9) s_base[tid] = cfg.base[tid];
    if (tid < ORBM_MAX_LEVELS) s_sf[tid] = F.sf[tid];
    // ---- the index (as collect_kernel: buckets of stream positions per octave pair, coarse column and fine row)
    uint32_t *cnt32 = lds;
    for (int i = tid; i < offn / 2; i += FUSED_THREADS) cnt32[i] = 0;
    for (int i = tid; i < fwords + 8 * 12 + 8 * 96; i += FUSED_THREADS) fbits[i] = 0;
    __syncthreads();
    const int T = (int)F.cell_off[ORBM_GRID_CELLS];
    const bool pairs = cfg.nstruct > 1;
    auto bucket_add = [&](int b) {
        const uint32_t old = atomicAdd(&cnt32[(b + 2) >> 1], ((b + 2) & 1) ? 0x10000u : 1u);
        return ((b + 2) & 1) ? old >> 16 : old & 0xFFFFu;
    };
    for (int j = tid; j < T; j += FUSED_THREADS) {
        const uint32_t w = __float_as_uint(F.cell_rec[j].w);
        const int oct = (int)((w >> 16) & 15u), cy = (int)((w >> 20) & 63u), cx = (int)(w >> 26);
        if (pairs) {
            if (oct <= 7) bucket_add(s_base[oct] + (cx >> s_k[oct]) * ORBM_GRID_ROWS + cy);
            if (oct + 1 <= 7) bucket_add(s_base[oct + 1] + (cx >> s_k[oct + 1]) * ORBM_GRID_ROWS + cy);
        } else {
            bucket_add((cx >> s_k[0]) * ORBM_GRID_ROWS + cy);
        }
    }
    __syncthreads();
    {
        const int per = (total + FUSED_THREADS - 1) / FUSED_THREADS;
        const int i0 = min(total, tid * per), i1 = min(total, i0 + per);
        uint32_t sum = 0;
        for (int i = i0; i < i1; i++) sum += c_off[i + 2];
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (lane == 63) s_part[wave] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (int w2 = 0; w2 < wave; w2++) run += s_part[w2];
        for (int i = i0; i < i1; i++) { const uint32_t c = c_off[i + 2]; c_off[i + 2] = (uint16_t)run; run += c; }
        if (tid == 0) { c_off[0] = 0; c_off[1] = 0; }
    }
    __syncthreads();
    const QSrc S = query_source<MODE>(P, inst);
    const int Q = S.Q;
    int32_t *out = P.out + (size_t)inst * P.stride;
    auto is_blocked = [&](uint32_t st, int dist) {
        if (MODE == MODE_PROJ || MODE == MODE_LAST) return (st & ORBM_MP_PRESENT) && (st & ORBM_MP_OBS);   // :113-115 / :1879-1881
        if (MODE == MODE_INIT) return (int)st <= dist;                                                     // :608
        return (st & ORBM_MP_PRESENT) != 0;                                                                // :2048 / :512
    };
    for (int j = tid; j < T; j += FUSED_THREADS) {
        const v4f rc = F.cell_rec[j];
        const uint32_t w = __float_as_uint(rc.w);
        const int oct = (int)((w >> 16) & 15u), cy = (int)((w >> 20) & 63u), cx = (int)(w >> 26);
        const int kp = (int)(w & 0xFFFFu);
        c_rec[j] = v4f{rc.x, rc.y, rc.z, __uint_as_float((uint32_t)j | ((uint32_t)oct << 13) | ((uint32_t)cx << 17) | ((uint32_t)cy << 23))};
        l_kp[j] = (uint16_t)kp;
        // slot state, by stream position
        uint32_t s0;
        if (MODE == MODE_INIT) s0 = (uint32_t)INT_MAX;                                   // vMatchedDistance (:573)
        else s0 = P.slot_flags ? P.slot_flags[(size_t)inst * P.stride + kp] : F.mp_flags[kp];
        state[j] = s0;
        owner[j] = -1;
        post[j] = 0;
        if (HIST_MODE) binmask[j] = 0;
        const bool is_free = MODE == MODE_INIT || !is_blocked(s0, 0);
        uint32_t p0 = 0xFFFFu, p1 = 0xFFFFu;
        if (pairs) {
            if (oct <= 7) p0 = bucket_add(s_base[oct] + (cx >> s_k[oct]) * ORBM_GRID_ROWS + cy);
            if (oct + 1 <= 7) p1 = bucket_add(s_base[oct + 1] + (cx >> s_k[oct + 1]) * ORBM_GRID_ROWS + cy);
        } else {
            p0 = bucket_add((cx >> s_k[0]) * ORBM_GRID_ROWS + cy);
        }
        const int blk = (cy >> 2) * 16 + (cx >> 2);
        const int s0i = pairs ? oct : 0, s1i = oct + 1;
        if (p0 != 0xFFFFu) {
            c_lst[p0] = (uint16_t)j;
            if (is_free) {
                atomicOr(&fbits[p0 >> 5], 1u << (p0 & 31));
                atomicAdd(&fcnt[s0i * 96 + (blk >> 1)], 1u << (16 * (blk & 1)));
                atomicOr(&fblk[s0i * 12 + (cy >> 2)], 1u << (cx >> 2));
            }
        }
        if (p1 != 0xFFFFu) {
            c_lst[p1] = (uint16_t)j;
            if (is_free) {
                atomicOr(&fbits[p1 >> 5], 1u << (p1 & 31));
                atomicAdd(&fcnt[s1i * 96 + (blk >> 1)], 1u << (16 * (blk & 1)));
                atomicOr(&fblk[s1i * 12 + (cy >> 2)], 1u << (cx >> 2));
            }
        }

    }
    if (MODE == MODE_INIT) for (int i = tid; i < Q; i += FUSED_THREADS) out[i] = -1;     // vnMatches12 (:566)
    if (MODE == MODE_INIT) {                                                             // the frame's descriptors, by stream position
        const gptr<const v4u> gd = (gptr<const v4u>)F.desc;
        for (int j = tid; j < 2 * T; j += FUSED_THREADS) {
            const int kp = (int)(__float_as_uint(F.cell_rec[j >> 1].w) & 0xFFFFu);
            ldesc[j] = gd[2 * kp + (j & 1)];
        }
    }
    if (tid < 36) s_hist[tid] = 0;
    if (tid < 3) s_any[tid] = 0;
    if (tid == 0) s_stamp_mem = 0;
    __syncthreads();
    {   // (counted after the scatter: its own pass, so that the counter's reset cannot race with the adds)
        int nf = 0;
        for (int j = tid; j < T; j += FUSED_THREADS) nf += (MODE == MODE_INIT || !is_blocked(state[j], 0)) ? 1 : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nf += __shfl_xor(nf, off, 64);
        if (tid == 0) s_free_total = 0;
        __syncthreads();
        if (lane == 0 && nf) atomicAdd(&s_free_total, nf);
        __syncthreads();
    }

    const bool use_hist = HIST_MODE && P.check_ori && (MODE == MODE_INIT || S.QS.angle != nullptr);
    const bool stereo_gate = has_stereo_gate<MODE>(S);
    const int thr = MODE == MODE_PROJ ? ORBM_TH_HIGH : (MODE == MODE_INIT ? ORBM_TH_LOW : P.dist_th);
    const gptr<const v4u> gdesc = (gptr<const v4u>)F.desc;
    const float min_x = F.min_x, min_y = F.min_y, inv_w = F.inv_w, inv_h = F.inv_h;
    auto wave_max = [&](int v) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
        return __builtin_amdgcn_readfirstlane(v);
    };
    // every LDS / global access of this wave issued so far has completed and none issued later has started
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
    };
    int my_matches = 0;
    // "does any thread of the workgroup ...?" with ONE barrier: three flags in rotation (as window_kernel)
    int any_phase = 0;
    auto wg_any = [&](bool pred) {
        const int cur = any_phase % 3;
        if (pred) s_any[cur] = 1;
        if (tid == 0) s_any[(any_phase + 1) % 3] = 0;
        lds_barrier();
        any_phase++;
        return s_any[cur] != 0;
    };
    // candidate words NW.. of a query, from its tail in global scratch (tcap is a multiple of 4)
    auto load_tail = [&](uint32_t (&cc)[WIN_CACHE], gptr<const uint32_t> tailp, int tc) {
        const gptr<const v4u> tq = (gptr<const v4u>)tailp;
#pragma unroll
        for (int g = 0; g < (WIN_CACHE - NW) / 4; g++)
            if (tc > 4 * g) {
                const v4u t4 = tq[g];
                cc[NW + 4 * g] = t4.x;
                cc[NW + 1 + 4 * g] = tc > 4 * g + 1 ? t4.y : W_INVALID;
                cc[NW + 2 + 4 * g] = tc > 4 * g + 2 ? t4.z : W_INVALID;
                cc[NW + 3 + 4 * g] = tc > 4 * g + 3 ? t4.w : W_INVALID;
            }
    };
    // The rounds (evaluate / contested? / decide, see the top of the file) over one query per lane.  WIDE: every wave of the
    // workgroup takes part (workgroup barriers between the phases, priority = position in the chunk); otherwise one wave on its
    // own (wave-synchronous).  A post is (round stamp << 10 | 1023 - priority) under atomicMax: a lower-priority-number query
    // of the same round wins, older rounds lose, the array is never cleared.  The stamp lives in LDS between calls.
    auto rounds = [&](auto wide_tag, bool unresolved, int prio, int qi, uint32_t qflags, float qangle, int ncand2, uint32_t (&cc)[WIN_CACHE],
                      gptr<const uint32_t> mytail) {
        constexpr bool WIDE = decltype(wide_tag)::value;
        auto sync = [&]() { if (WIDE) lds_barrier(); else wave_sync(); };
        auto any = [&](bool p) { return WIDE ? wg_any(p) : (bool)__any(p); };
        uint32_t rstamp = *s_stamp;                          // uniform: the last writer finished before the barrier that let us in
        const int wave_ncand = wave_max(ncand2);
        while (any(unresolved)) {
            rstamp++;
            const uint32_t mypost = (rstamp << 10) | (uint32_t)(1023 - prio);
            auto lower_posted = [&](int k) {                 // has a lower-index unresolved query posted on slot k in this round?
                const uint32_t v = post[k];
                return (v >> 10) == rstamp && (1023u - (v & 1023u)) < (uint32_t)prio;
            };
            uint32_t bestW = W_INVALID, secondW = W_INVALID;
            uint32_t freem = 0;                              // candidates (of the first WIN_CACHE) found free in this round
            auto consider = [&](uint32_t w, uint32_t stv) {
                const int k = (int)word_slot(w), dist = word_dist(w);
                if (is_blocked(stv, dist)) return false;
                if (dist <= thr) atomicMax(&post[k], mypost);                              // this query may still claim k
                secondW = min(max(bestW, w), secondW);
                bestW = min(bestW, w);
                return true;
            };
            // ---- evaluate: the slot states of four candidates are requested together
            if (unresolved) {
                {
#pragma unroll
                    for (int g = 0; g < WIN_CACHE / 4; g++) {
                        if (wave_ncand > 4 * g) {
                            uint32_t stv[4];
#pragma unroll
                            for (int u = 0; u < 4; u++) stv[u] = 4 * g + u < ncand2 ? state[word_slot(cc[4 * g + u])] : 0u;
#pragma unroll
                            for (int u = 0; u < 4; u++)
                                if (4 * g + u < ncand2 && consider(cc[4 * g + u], stv[u])) freem |= 1u << (4 * g + u);
                        }
                    }
                    // candidates beyond the registers (very wide windows): four words per load from the query's tail
                    for (int i = WIN_CACHE; i < wave_ncand; i += 4)
                        if (i < ncand2) {
                            const v4u t4 = *(gptr<const v4u>)(mytail + (i - NW));
                            const uint32_t tw[4] = {t4.x, t4.y, t4.z, t4.w};
                            uint32_t stv[4];
#pragma unroll
                            for (int u = 0; u < 4; u++) stv[u] = i + u < ncand2 ? state[word_slot(tw[u])] : 0u;
#pragma unroll
                            for (int u = 0; u < 4; u++) if (i + u < ncand2) consider(tw[u], stv[u]);
                        }
                }
            }
            const int none = MODE == MODE_INIT ? INT_MAX : 256;
            const bool has1 = word_valid(bestW), has2 = word_valid(secondW);
            const int bestDist = has1 ? word_dist(bestW) : none, bestDist2 = has2 ? word_dist(secondW) : none;
            const int bestIdx = has1 ? (int)word_slot(bestW) : -1, bestIdx2 = has2 ? (int)word_slot(secondW) : -1;
            const int bestLevel = has1 ? word_oct(bestW) : -1, bestLevel2 = has2 ? word_oct(secondW) : -1;
            const bool alive = unresolved && bestIdx >= 0 && bestDist <= thr;
            // The second-best free slot enters the outcome through the ratio test only, and slots only ever become blocked, so
            // its distance can only grow: a test that lets the query through now (bestDist <= ratio * bestDist2 for a6 --
            // whatever the two levels are --, bestDist < bestDist2 * ratio for a5) lets it through with any later second best
            // as well.  Only a query whose test does NOT pass on distances now depends on that slot staying free: it names it,
            // so that no later query takes it first, and waits if a lower query may take it.
            bool second_matters = false;
            if (USES_SECOND && alive && bestIdx2 >= 0) {
                if (MODE == MODE_PROJ) second_matters = (float)bestDist > __fmul_rn(P.ratio, (float)bestDist2);
                else second_matters = !((float)bestDist < __fmul_rn((float)bestDist2, P.ratio));
            }
            if (second_matters) atomicMax(&post[bestIdx2], mypost);
            sync();
            // ---- contested?
            bool contested = false;
            if (alive) {
                contested = lower_posted(bestIdx);
                if (second_matters && lower_posted(bestIdx2)) contested = true;
            }
            if (any(contested)) {
                if (contested) {
                    {
#pragma unroll
                        for (int i = 0; i < WIN_CACHE; i++)
                            if ((freem >> i) & 1u) atomicMax(&post[word_slot(cc[i])], mypost);
                        for (int i = WIN_CACHE; i < wave_ncand; i += 4)
                            if (i < ncand2) {
                                const v4u t4 = *(gptr<const v4u>)(mytail + (i - NW));
                                const uint32_t tw[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
                                for (int u = 0; u < 4; u++)
                                    if (i + u < ncand2) {
                                        const int k = (int)word_slot(tw[u]);
                                        if (!is_blocked(state[k], word_dist(tw[u]))) atomicMax(&post[k], mypost);
                                    }
                            }
                    }
                }
                sync();
            }
            // ---- decide
            if (unresolved) {
                bool is_final = !alive;
                if (alive && !contested) {
                    bool accept0;
                    if (MODE == MODE_PROJ) accept0 = !(bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(P.ratio, (float)bestDist2));
                    else if (MODE == MODE_INIT) accept0 = (float)bestDist < __fmul_rn((float)bestDist2, P.ratio);
                    else accept0 = true;
                    is_final = !accept0 || !lower_posted(bestIdx);
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
                        const int bestKp = (int)l_kp[bestIdx];
                        if (MODE == MODE_INIT) {
                            const int old = owner[bestIdx];
                            if (old >= 0) { out[old] = -1; my_matches--; }                 // :629-633
                            out[qi] = bestKp; owner[bestIdx] = qi; state[bestIdx] = (uint32_t)bestDist;  // :636-638
                        } else {
                            uint32_t ns = ORBM_MP_PRESENT;
                            if ((MODE != MODE_PROJ && MODE != MODE_LAST) || (qflags & ORBM_Q_OBS)) ns |= ORBM_MP_OBS;
                            state[bestIdx] = ns;
                            owner[bestIdx] = qi;                                           // :160 / :1906 / :2066 / :535
                            if (is_blocked(ns, 0)) {                                       // the searching waves stop seeing the record
                                // where the record sits in the lists of its (up to two) structures: found by walking its bucket's
                                // short run (a claim is rare next to a search; a table of positions would cost 4 B of LDS per keypoint)
                                const uint32_t rw = __float_as_uint(c_rec[bestIdx].w);
                                const int roct = (int)((rw >> 13) & 15u), rcx = (int)((rw >> 17) & 63u), rcy = (int)((rw >> 23) & 63u);
                                const int blk = (rcy >> 2) * 16 + (rcx >> 2);
                                auto unfree = [&](int si) {
                                    const int b = s_base[si] + (rcx >> s_k[si]) * ORBM_GRID_ROWS + rcy;
                                    uint32_t pp = c_off[b + 1];
                                    const uint32_t pe = c_off[b + 2];
                                    while (pp < pe && c_lst[pp] != (uint16_t)bestIdx) pp++;
                                    if (pp < pe) atomicAnd(&fbits[pp >> 5], ~(1u << (pp & 31)));
                                    const uint32_t old = atomicSub(&fcnt[si * 96 + (blk >> 1)], 1u << (16 * (blk & 1)));
                                    if (((old >> (16 * (blk & 1))) & 0xFFFFu) == 1u) atomicAnd(&fblk[si * 12 + (rcy >> 2)], ~(1u << (rcx >> 2)));
                                };
                                if (!pairs) unfree(0);
                                else {
                                    if (roct <= 7) unfree(roct);
                                    if (roct + 1 <= 7) unfree(roct + 1);
                                }
                                atomicSub(&s_free_total, 1);
                            }
                        }
                        my_matches++;
                        if (use_hist) {
                            const int bin = rot_bin(qangle, F.angle[bestKp]);              // :644-651 / :1911-1918 / :2079-2090
                            atomicAdd(&hist[bin], 1);
                            if (MODE != MODE_INIT) binmask[bestIdx] |= 1u << bin;
                        }
                    }
                }
            }
            if (!WIDE) sync();      // (WIDE: the barrier of the loop condition orders the slot writes before the next round's reads)
        }
        // rstamp is uniform over the participants.  WIDE: everybody has passed the barrier of the last loop condition, and nobody
        // reads the stamp again before the next chunk's barrier (B)
        if (WIDE ? tid == 0 : lane == 0) *s_stamp = rstamp;
    };

#ifdef ORBM_DIAG
    if (trace && inst == 0 && tid == 0) trace[1] = wall_clock64();
#endif
    const int n_chunks = (Q + FUSED_CHUNK - 1) / FUSED_CHUNK;
    const int tid_c = tid - 64;                             // waves 1..15: the thread's query of a chunk
    // The parameters of a query are requested PF chunks before its chunk is searched: a chunk takes about a microsecond once
    // the frame's slots have filled up, a load from HBM three
    constexpr int PF = 2;
    RawQ rawq[PF];
#pragma unroll
    for (int d = 0; d < PF; d++) rawq[d] = raw_load<MODE>(P, S, wave > 0 ? d * FUSED_CHUNK + tid_c : Q);
    for (int c = 0; c < n_chunks; c++) {
        const int base_q = c * FUSED_CHUNK;
        uint32_t *mytail_w = tail_scratch + ((size_t)(inst * 2 + (c & 1)) * FUSED_CHUNK + max(tid_c, 0)) * tcap;
        uint32_t kw[NW];
#pragma unroll
        for (int i = 0; i < NW; i++) kw[i] = W_INVALID;
        int ncand = 0;
        uint32_t mind = 0xFFFFu;
        uint32_t meta_bits = 0, angle_bits = 0;
        if (wave > 0) {
            // ================= search: one query per lane =================
            const int qi = base_q + tid_c;
            Query q = raw_decode<MODE>(P, rawq[0], s_sf);
#pragma unroll
            for (int d = 0; d + 1 < PF; d++) rawq[d] = rawq[d + 1];
            rawq[PF - 1] = raw_load<MODE>(P, S, qi + PF * FUSED_CHUNK);
            const float xr = __fsub_rn(q.x, min_x), yr = __fsub_rn(q.y, min_y), r = q.r;
            const int x0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(xr, r), inv_w)));
            const int x1 = min(ORBM_GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(xr, r), inv_w)));
            const int y0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(yr, r), inv_h)));
            const int y1 = min(ORBM_GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(yr, r), inv_h)));
            int lo = -1, hi = 255;
            if (MODE == MODE_SIM3) { lo = q.minL; hi = q.maxL; }
            else if (q.minL > 0 || q.maxL >= 0) { lo = q.minL; if (q.maxL >= 0) hi = q.maxL; }
            meta_bits = (q.flags & ORBM_Q_OBS) ? 0x80000000u : 0u;
            angle_bits = __float_as_uint(q.angle);
            bool uncovered = false;
            int st = 0;
            bool covered = !pairs;
            if (pairs && hi <= 7 && hi >= 0 && hi - lo <= 1) { st = hi; covered = true; }
            bool more = q.valid && x0 < ORBM_GRID_COLS && x1 >= 0 && y0 < ORBM_GRID_ROWS && y1 >= 0 && x1 >= x0 && y1 >= y0;
            if (more && !covered) { uncovered = true; more = false; }
            if (MODE != MODE_INIT && more) {
                // is there a free record of this octave pair in any 4 x 4-cell block the window touches?  (four block rows
                // at most are looked at; a taller window walks)
                const int BY0 = y0 >> 2, BY1 = y1 >> 2;
                const uint32_t colmask = ((2u << (x1 >> 2)) - 1u) & ~((1u << (x0 >> 2)) - 1u);
                const uint32_t *fr = fblk + st * 12 + BY0;
                uint32_t rows = fr[0];
                if (BY1 >= BY0 + 1) rows |= fr[1];
                if (BY1 >= BY0 + 2) rows |= fr[2];
                if (BY1 >= BY0 + 3) rows |= fr[3];
                if (BY1 <= BY0 + 3 && !(rows & colmask)) more = false;
            }
            if constexpr (MODE == MODE_INIT) {
                // a5: the query only publishes its window and its descriptor; the waves search the chunk's queries together below
                uint32_t win = 0xFFFFFFFFu;
                if (more && lo == hi) {
                    load_desc<MODE>(S, qi, q);
                    s_qdesc[2 * tid_c] = q.d0; s_qdesc[2 * tid_c + 1] = q.d1;
                    s_qx[tid_c] = q.x; s_qy[tid_c] = q.y;
                    win = (uint32_t)x0 | ((uint32_t)x1 << 6) | ((uint32_t)y0 << 12) | ((uint32_t)y1 << 18) | ((uint32_t)st << 24) | ((uint32_t)hi << 27);
                } else if (more) uncovered = true;
                s_qwin[tid_c] = win;
                if (uncovered) *redo_flag = 1;
            } else {
            const int kk = s_k[st], sb = s_base[st];
            int X = (x0 >> kk) - 1;
            const int X1 = x1 >> kk;
            uint32_t j = 0, e = 0, fm = 0, jb = 0;          // list positions [j, e) of the column still to look at; fm = free records of [jb, jb + 32)
            uint32_t park01 = 0;                            // FPARK stream positions, 16 bits each
            int np = 0;
            // the query's own descriptor is requested now (it travels while the window is walked) -- only for queries whose
            // window holds a free record at all
            bool have_desc = false;
            uint32_t qd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (more) {
                load_desc<MODE>(S, qi, q);
                qd[0] = q.d0.x; qd[1] = q.d0.y; qd[2] = q.d0.z; qd[3] = q.d0.w; qd[4] = q.d1.x; qd[5] = q.d1.y; qd[6] = q.d1.z; qd[7] = q.d1.w;
                have_desc = true;
            }
            // distances of the parked survivors: the loads of all of them are requested together
            auto flush = [&]() {
                if (np > 0 && !have_desc) {
                    load_desc<MODE>(S, qi, q);
                    qd[0] = q.d0.x; qd[1] = q.d0.y; qd[2] = q.d0.z; qd[3] = q.d0.w; qd[4] = q.d1.x; qd[5] = q.d1.y; qd[6] = q.d1.z; qd[7] = q.d1.w;
                    have_desc = true;
                }
                v4u dlo[FPARK], dhi[FPARK];
                uint32_t ppos[FPARK], pw[FPARK];
#pragma unroll
                for (int i = 0; i < FPARK; i++) {
                    ppos[i] = (i & 1) ? park01 >> 16 : park01 & 0xFFFFu;
                    dlo[i] = dhi[i] = v4u{0, 0, 0, 0}; pw[i] = 0;
                    if (i < np) {
                        pw[i] = __float_as_uint(c_rec[ppos[i]].w);
                        const int kp = (int)l_kp[ppos[i]];
                        dlo[i] = gdesc[2 * kp]; dhi[i] = gdesc[2 * kp + 1];
                    }
                }
#pragma unroll
                for (int i = 0; i < FPARK; i++) {
                    if (i < np) {
                        const uint32_t cd[8] = {dlo[i].x, dlo[i].y, dlo[i].z, dlo[i].w, dhi[i].x, dhi[i].y, dhi[i].z, dhi[i].w};
                        const uint32_t dist = hamming256(qd, cd);
                        bool keepit = true;
                        if (MODE == MODE_INIT) keepit = !((int)state[ppos[i]] <= (int)dist);        // :608, with the distance
                        if (keepit) {
                            const uint32_t x = (dist << 17) | (ppos[i] << 4) | ((pw[i] >> 13) & 15u);
#pragma unroll
                            for (int s2 = 0; s2 < NW; s2++) if (ncand == s2) kw[s2] = x;
                            if (ncand >= NW && ncand < tcap + NW) mytail_w[ncand - NW] = x;
                            mind = min(mind, dist);
                            ncand++;
                        }
                    }
                }
                np = 0; park01 = 0;
            };
            for (;;) {
                if (more) {
                    if (fm == 0) {
                        if (j >= e) {                       // next coarse column of the window
                            if (++X > X1) more = false;
                            else {
                                const int b = sb + X * ORBM_GRID_ROWS;
                                j = c_off[b + y0 + 1];
                                e = c_off[b + y1 + 2];
                            }
                        }
                        if (more && j < e) {                // the free records among the next 32 list positions
                            const uint32_t w0 = fbits[j >> 5], w1 = fbits[(j >> 5) + 1];
                            const uint32_t win = (uint32_t)((((uint64_t)w1 << 32) | w0) >> (j & 31));
                            const uint32_t len = e - j;
                            fm = len >= 32 ? win : win & ((1u << len) - 1u);
                            jb = j; j += 32;
                        }
                    }
                    if (fm != 0) {
                        const int bit = __ffs((int)fm) - 1;
                        fm &= fm - 1;
                        const uint32_t jcur = c_lst[jb + bit];
                        const v4f rc = c_rec[jcur];
                        const uint32_t w = __float_as_uint(rc.w);
                        const int oct = (int)((w >> 13) & 15u), cx = (int)((w >> 17) & 63u);
                        int pass = (int)(cx >= x0) & (int)(cx <= x1) & (int)(oct >= lo) & (int)(oct <= hi) &
                                   (int)(fabsf(__fsub_rn(rc.x, q.x)) < r) & (int)(fabsf(__fsub_rn(rc.y, q.y)) < r);
                        if (stereo_gate) pass &= 1 ^ ((int)(rc.z > 0) & (int)(fabsf(__fsub_rn(q.xr, rc.z)) > r));   // :117-122 / :1883-1891
                        if (pass) {
                            park01 |= jcur << (16 * np);
                            np++;
                        }
                    }
                }
                if (__any(np == FPARK)) flush();
                if (!__any(more)) break;
            }
            if (__any(np > 0)) flush();
            // more candidates than this form has words for (or a level range the index does not cover: cannot happen with the
            // launch rules): the instance is flagged and searched again, whole, by the general kernel that follows this one
            if (uncovered || ncand > tcap + NW) { *redo_flag = 1; ncand = min(ncand, tcap + NW); }
            }
        }
        if constexpr (MODE == MODE_INIT) {
            // ---- a5: cooperative search.  Its valid queries are the level-0 keypoints of frame 1, a few hundred of them side by
            // side in the query order (a handful of waves), each with some thirty candidates in a 2 x 100 px window and a
            // distance to compute for every one (the slot test `vMatchedDistance[i2] <= dist`, :608, needs it): one query per
            // lane made that a chain of thirty dependent steps on four waves while eleven idled.  Here wave w takes the
            // queries w - 1, w + 14, ... of the chunk one after the other and spreads each one's candidates over its lanes:
            // the runs of the window's coarse columns are laid end to end (a prefix sum over the columns, one per lane), lane
            // l takes entry l, applies the gates, computes its distance against descriptors held in LDS, and the words of
            // the survivors go, packed, to the query's tail in global scratch (ALL words: the owner reads its first eight
            // back).  The order of the words does not matter: they carry the reference's ranking themselves.
            lds_barrier();                                  // (I1) the chunk's windows and descriptors are in LDS
            if (wave > 0) {
                const float r = (float)P.window;
                for (int t = 0; t < 64; t++) {
                    const int pos = (wave - 1) + 15 * t;
                    const uint32_t win = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_qwin[pos]);
                    if (win == 0xFFFFFFFFu) continue;
                    const int x0 = (int)(win & 63u), x1 = (int)((win >> 6) & 63u), y0 = (int)((win >> 12) & 63u), y1 = (int)((win >> 18) & 63u);
                    const int st = (int)((win >> 24) & 7u), lv = (int)((win >> 27) & 15u);
                    const float qx = s_qx[pos], qy = s_qy[pos];
                    const v4u q0 = s_qdesc[2 * pos], q1 = s_qdesc[2 * pos + 1];
                    const uint32_t qd[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                    const int kk = s_k[st], sb = s_base[st];
                    const int X0 = x0 >> kk, ncol = (x1 >> kk) - X0 + 1;
                    uint32_t jc = 0, len = 0;
                    if (lane < ncol) {
                        const int b = sb + (X0 + lane) * ORBM_GRID_ROWS;
                        jc = c_off[b + y0 + 1];
                        len = (uint32_t)c_off[b + y1 + 2] - jc;
                    }
                    uint32_t incl = len;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        const uint32_t up = __shfl_up(incl, d, 64);
                        if (lane >= d) incl += up;
                    }
                    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    uint32_t *tailq = tail_scratch + ((size_t)(inst * 2 + (c & 1)) * FUSED_CHUNK + pos) * tcap;
                    uint32_t cnt = 0, md = 0xFFFFu;
                    for (uint32_t eb = 0; eb < total; eb += 64) {
                        const uint32_t l = eb + (uint32_t)lane;
                        const bool have = l < total;
                        uint32_t idx = 0;
                        for (int cc2 = 0; cc2 < ncol; cc2++) {
                            const uint32_t ic = (uint32_t)__builtin_amdgcn_readlane((int)incl, cc2), lc = (uint32_t)__builtin_amdgcn_readlane((int)len, cc2);
                            const uint32_t j0 = (uint32_t)__builtin_amdgcn_readlane((int)jc, cc2);
                            if (l < ic && l >= ic - lc) idx = j0 + (l - (ic - lc));
                        }
                        bool keep = false;
                        uint32_t x = 0, dist = 0xFFFFu;
                        if (have) {
                            const uint32_t jcur = c_lst[idx];
                            const v4f rc = c_rec[jcur];
                            const uint32_t w = __float_as_uint(rc.w);
                            const int oct = (int)((w >> 13) & 15u), cx = (int)((w >> 17) & 63u);
                            const int pass = (int)(cx >= x0) & (int)(cx <= x1) & (int)(oct == lv) &
                                             (int)(fabsf(__fsub_rn(rc.x, qx)) < r) & (int)(fabsf(__fsub_rn(rc.y, qy)) < r);
                            if (pass) {
                                const v4u dlo = ldesc[2 * jcur], dhi = ldesc[2 * jcur + 1];
                                const uint32_t cd[8] = {dlo.x, dlo.y, dlo.z, dlo.w, dhi.x, dhi.y, dhi.z, dhi.w};
                                dist = hamming256(qd, cd);
                                keep = !((int)state[jcur] <= (int)dist);                      // :608, with the distance
                                x = (dist << 17) | (jcur << 4) | (uint32_t)oct;
                            }
                        }
                        const unsigned long long km = __ballot(keep);
                        if (keep) {
                            const uint32_t pp = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
                            if (pp < (uint32_t)tcap) tailq[pp] = x;
                            md = min(md, dist);
                        }
                        cnt += (uint32_t)__popcll(km);
                    }
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) md = min(md, (uint32_t)__shfl_xor((int)md, off, 64));
                    if (cnt > (uint32_t)tcap) { *redo_flag = 1; cnt = (uint32_t)tcap; }
                    if (lane == 0) s_qres[pos] = cnt | (md << 16);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // the tail words are global memory
            }
            lds_barrier();                                  // (I2) every query's words and count are there
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (wave > 0 && s_qwin[tid_c] != 0xFFFFFFFFu) {
                const uint32_t res = s_qres[tid_c];
                ncand = (int)(res & 0xFFFFu); mind = res >> 16;
                if (ncand > 0) {
                    const gptr<const v4u> tq = (gptr<const v4u>)in_global(mytail_w);
                    const v4u a = tq[0];
                    kw[0] = a.x; kw[1] = ncand > 1 ? a.y : W_INVALID; kw[2] = ncand > 2 ? a.z : W_INVALID; kw[3] = ncand > 3 ? a.w : W_INVALID;
                    if (ncand > 4) {
                        const v4u b4 = tq[1];
                        kw[4] = b4.x; kw[5] = ncand > 5 ? b4.y : W_INVALID; kw[6] = ncand > 6 ? b4.z : W_INVALID; kw[7] = ncand > 7 ? b4.w : W_INVALID;
                    }
                }
            }
        }
        const bool live = wave > 0 && ncand > 0 && (int)mind <= thr;
        const unsigned long long lmask = __ballot(live);
        // tail words (global memory) written by this wave must have landed before another wave is told about them; a wave that
        // wrote none keeps its outstanding loads in flight
        if (__any(live && ncand > NW)) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (wave > 0 && lane == 0) s_live[wave - 1] = lmask;
        lds_barrier();                                      // (B) every wave's live mask is there; wave 0 has settled chunk c - 1
#ifdef ORBM_DIAG
        if (trace && inst == 0 && tid == 0 && c < 190) trace[800 + c] = wall_clock64();
#endif
        int L, before;                                      // live queries of the chunk; ... of the waves before this one
        {
            const int cw = lane < 15 ? (int)__popcll(s_live[lane]) : 0;
            int incl = cw;
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                const int up = __shfl_up(incl, d, 64);
                if (lane >= d) incl += up;
            }
            L = __builtin_amdgcn_readlane(incl, 14);
            before = wave >= 2 ? __shfl(incl, wave - 2, 64) : 0;
        }
        // no free slot is left in the frame and nothing is pending: every remaining query would skip every candidate
        // (ORBmatcher.cc:113-115): the search is over
        if (L == 0 && s_free_total == 0) break;
        // (a5: the tail holds ALL of a query's words, the code below expects the ninth at its start)
        const gptr<const uint32_t> mytail_r = in_global(mytail_w) + (MODE == MODE_INIT ? NW : 0);
        if (L > NARROW_MAX) {
            // ---- many live queries (the first chunks of an instance): every searching lane keeps its own query and the whole
            // workgroup goes through the rounds together, claim priority = position in the chunk
            uint32_t cc[WIN_CACHE];
#pragma unroll
            for (int i = 0; i < WIN_CACHE; i++) cc[i] = i < NW ? kw[i] : W_INVALID;
            if (live && ncand > NW) load_tail(cc, mytail_r, ncand - NW);
            rounds(std::true_type{}, live, max(tid_c, 0), base_q + tid_c, (meta_bits & 0x80000000u) ? ORBM_Q_OBS : 0u, __uint_as_float(angle_bits),
                   live ? ncand : 0, cc, mytail_r);
#ifdef ORBM_DIAG
            if (trace && inst == 0 && tid == 0 && c < 250) { trace[2 + 2 * c] = wall_clock64(); trace[3 + 2 * c] = trace[2 + 2 * c]; trace[600 + c] = (unsigned long long)L; }
#endif
            continue;
        }
        // ---- few live queries: they go to wave 0 through the pending list, and the next chunk is searched meanwhile
        if (live) {
            const int rk = before + (int)__popcll(lmask & ((1ull << lane) - 1ull));
            uint32_t *pe = pending + (size_t)rk * PEND_WORDS;
            pe[0] = meta_bits | ((uint32_t)min(ncand, 0xFFFF) << 10) | (uint32_t)tid_c;
            pe[1] = angle_bits;
#pragma unroll
            for (int i = 0; i < NW; i++) pe[2 + i] = kw[i];
        }
        lds_barrier();                                      // (A) the pending list of chunk c is complete
#ifdef ORBM_DIAG
        if (trace && inst == 0 && tid == 0 && c < 250) trace[2 + 2 * c] = wall_clock64();
#endif
        if (wave != 0) continue;
        if (L > 0) {
            const bool mine = lane < L;
            uint32_t cc[WIN_CACHE];
#pragma unroll
            for (int i = 0; i < WIN_CACHE; i++) cc[i] = W_INVALID;
            uint32_t meta = 0, abits = 0;
            if (mine) {
                const uint32_t *pe = pending + (size_t)lane * PEND_WORDS;
                meta = pe[0]; abits = pe[1];
#pragma unroll
                for (int i = 0; i < NW; i++) cc[i] = pe[2 + i];
            }
            const int t_c = (int)(meta & 1023u);
            const int ncand2 = mine ? (int)((meta >> 10) & 0xFFFFu) : 0;
            const gptr<const uint32_t> mytail = in_global(tail_scratch) + ((size_t)(inst * 2 + (c & 1)) * FUSED_CHUNK + t_c) * tcap + (MODE == MODE_INIT ? NW : 0);
            if (__any(mine && ncand2 > NW)) {
                if (mine && ncand2 > NW) load_tail(cc, mytail, ncand2 - NW);
            }
            rounds(std::false_type{}, mine, lane, base_q + t_c, (meta & 0x80000000u) ? ORBM_Q_OBS : 0u, __uint_as_float(abits), ncand2, cc, mytail);
        }
#ifdef ORBM_DIAG
        if (trace && inst == 0 && tid == 0 && c < 250) { trace[3 + 2 * c] = wall_clock64(); trace[600 + c] = (unsigned long long)L; }
#endif
    }
#ifdef ORBM_DIAG
    if (trace && inst == 0 && tid == 0) trace[599] = wall_clock64();
#endif
    if (my_matches) atomicAdd(nmatch, my_matches);
    __syncthreads();
    // a flagged instance is searched again from the beginning by the kernel that follows: nothing of this attempt may reach
    // the outputs (prev_xy of SearchForInitialization is an input of that pass as well)
    if (*redo_flag != 0) return;
    if (use_hist) {
        if (tid == 0) {
            int i1, i2, i3;
            three_maxima(hist, i1, i2, i3);
            keep[0] = i1; keep[1] = i2; keep[2] = i3;
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
            for (int i = tid; i < Q; i += FUSED_THREADS) {
                const int m = out[i];
                if (m < 0) continue;                                                      // :673 (a stolen match is already -1)
                const int bin = rot_bin(S.F1.angle[i], F.angle[m]);
                if (!((keepmask >> bin) & 1u)) { out[i] = -1; removed++; }
            }
            if (removed) atomicSub(nmatch, removed);
        } else {
            for (int i = tid; i < T; i += FUSED_THREADS)
                if (binmask[i] & ~keepmask) owner[i] = -2;                                 // slot set to NULL (:1942 / :2115)
        }
        __syncthreads();
    }
    if (MODE == MODE_INIT) {
        for (int i = tid; i < Q; i += FUSED_THREADS) {                                    // :684-686
            const int m = out[i];
            if (m >= 0) { P.prev_out[2 * i] = F.x[m]; P.prev_out[2 * i + 1] = F.y[m]; }
        }
    } else {
        for (int i = tid; i < n; i += FUSED_THREADS) out[i] = -1;                         // keypoints outside the grid are never candidates
        __syncthreads();
        for (int i = tid; i < T; i += FUSED_THREADS) out[l_kp[i]] = owner[i];
    }
    if (tid == 0) P.counts[inst] = *nmatch;
}

// ---- the claim loop.  PRE = false: collects inline (one kernel does everything; frame optionally staged in LDS).
// PRE = true: kernel 2 of 2 of the split form, candidates come from collect_kernel's rows.
template <int MODE, bool LDS_FRAME, bool PRE>
__global__ void window_kernel(WinParams P, int waves_per_inst, const v4u *head_in, const uint32_t *tail_in, int tcap, const int32_t *redo_flags) {
    extern __shared__ __align__(16) uint32_t lds[];
    __shared__ int s_any[3];
    const int inst = blockIdx.x;
    // as the pass after fused_kernel: only the instances that kernel flagged (a query with more candidates than it holds) are
    // searched, from the beginning; the others leave at once
    if (redo_flags && redo_flags[inst] == 0) return;
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

    // Slots: a keypoint is named by its position j in the cell-ordered stream: that is what ranks candidates of equal
    // distance (the reference's visit order), so candidate words order themselves.
    const int n_slots = (int)F.cell_off[ORBM_GRID_CELLS];
    if (PRE) WIN_STAMP(16);
    for (int i = tid; i < n_slots; i += nthr) {
        uint32_t s0;
        const int kp = (int)F.cell_idx[i];
        if (MODE == MODE_INIT) s0 = (uint32_t)INT_MAX;                                   // vMatchedDistance (:573)
        else s0 = P.slot_flags ? P.slot_flags[(size_t)inst * P.stride + kp] : F.mp_flags[kp];
        state[i] = s0;
        owner[i] = -1;
        post0[i] = NOPOST; post1[i] = NOPOST;
        binmask[i] = 0;
    }
    if (MODE == MODE_INIT) for (int i = tid; i < Q; i += nthr) out[i] = -1;              // vnMatches12 (:566)
    if (tid < 32) hist[tid] = 0;
    if (tid == 0) *nmatch = 0;
    if (tid < 3) s_any[tid] = 0;            // wg_any's flags (below)
    __syncthreads();
    if (PRE) WIN_STAMP(17);

    const bool use_hist = P.check_ori && (MODE == MODE_LAST || MODE == MODE_KF || MODE == MODE_INIT) &&
                          (MODE == MODE_INIT || S.QS.angle != nullptr);
    const bool stereo_gate = has_stereo_gate<MODE>(S);
    const int thr = MODE == MODE_PROJ ? ORBM_TH_HIGH : (MODE == MODE_INIT ? ORBM_TH_LOW : P.dist_th);
    int my_matches = 0;
    const gptr<const v4u> heads = PRE ? (gptr<const v4u>)head_in + (size_t)inst * waves_per_inst * 64 : nullptr;
    const gptr<const uint32_t> tails = PRE ? in_global(tail_in) + (size_t)inst * waves_per_inst * 64 * tcap : nullptr;

    auto load_query = [&](int qi, bool rewalk) {
        Query q = load_params<MODE>(P, F, S, qi);
        if (PRE) {
            if (rewalk && q.valid) load_desc<MODE>(S, qi, q);  // only the re-walking fallback compares descriptors here
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
    // "does any thread of the workgroup ...?" with ONE barrier (__syncthreads_or is three): three flags in rotation; the one
    // used next is cleared by thread 0 before this call's barrier (its last readers passed the previous barrier already)
    int any_phase = 0;
    auto wg_any = [&](bool pred) {
        const int cur = any_phase % 3;
        if (pred) s_any[cur] = 1;
        if (tid == 0) s_any[(any_phase + 1) % 3] = 0;
        __syncthreads();
        any_phase++;
        return s_any[cur] != 0;
    };
    auto is_blocked = [&](uint32_t st, int dist) {
        if (MODE == MODE_PROJ || MODE == MODE_LAST) return (st & ORBM_MP_PRESENT) && (st & ORBM_MP_OBS);   // :113-115 / :1879-1881
        if (MODE == MODE_INIT) return (int)st <= dist;                                                     // :608
        return (st & ORBM_MP_PRESENT) != 0;                                                                // :2048 / :512
    };

    // One chunk of up to blockDim queries, one per lane, in ascending query index over tid (claim priority = tid).
    // Called by every thread of the workgroup (barriers inside).
    // cc: the query's candidates as words dist << 17 | slot << 4 | octave (ascending = the reference's ranking); `complete`
    // = cc holds every candidate (else the rounds walk the window again)
    auto process_chunk = [&](const int qi, const Query &cur, const int nc_pre, const bool complete, uint32_t (&cc)[WIN_CACHE]) {
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
            auto collect = [&](int k, int oct, float ur, int j) {
                if (!static_gates<MODE>(cur, stereo_gate, oct, ur)) return;
                if (ncand < WIN_CACHE) {
                    const v4u lo = acc.desc[2 * k], hi = acc.desc[2 * k + 1];
                    const uint32_t cd[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    const uint32_t v = ((uint32_t)j << 4) | (uint32_t)oct | (hamming256(qd, cd) << 17);
#pragma unroll
                    for (int i = 0; i < WIN_CACHE; i++) if (i == ncand) cc[i] = v;
                }
                ncand++;
            };
            if (MODE == MODE_SIM3) walk_window<true>(acc, x, y, r, minL, maxL, collect);
            else walk_window<false>(acc, x, y, r, minL, maxL, collect);
        }
        if (ncand == 0) unresolved = false;   // empty window: nothing to match, nothing to claim (:91-92)
        // candidate i of this lane's query: the first WIN_CACHE in registers, the rest (split form) re-read from the
        // query's tail; i is wave-uniform
        const bool cached = PRE ? complete : ncand <= WIN_CACHE;
        const int wave_ncand = wave_max(cached ? ncand : 0);
        const gptr<const uint32_t> mytail = PRE ? tails + (size_t)min(qi, max(Q - 1, 0)) * tcap : nullptr;
        auto cand_at = [&](int i) -> uint32_t {
            if (i < WIN_CACHE) return cc[i];
            return (PRE && i < ncand) ? mytail[i - 4] : W_INVALID;
        };

        // A query none of whose claimable candidates (distance <= threshold) is still free can never be accepted, and
        // blocked slots stay blocked, so it is settled without taking part in the rounds; a chunk with no live query
        // costs one barrier.
        if (cached && unresolved) {
            bool live = false;
            for (int i = 0; i < wave_ncand; i++) {
                const uint32_t v = cand_at(i);
                if (i < ncand && word_dist(v) <= thr) {
                    const uint32_t st = state[word_slot(v)];
                    bool blocked;
                    if (MODE == MODE_PROJ || MODE == MODE_LAST) blocked = (st & ORBM_MP_PRESENT) && (st & ORBM_MP_OBS);
                    else if (MODE == MODE_INIT) blocked = (int)st <= word_dist(v);
                    else blocked = (st & ORBM_MP_PRESENT) != 0;
                    if (!blocked) live = true;
                }
            }
            unresolved = live;
        }
        if (!wg_any(unresolved)) return;

        int round = 0;
        if (PRE) WIN_STAMP(20);
        while (true) {
            uint32_t *post = (round & 1) ? post1 : post0;
            uint32_t *post_next = (round & 1) ? post0 : post1;
            // The reference's inner loop body for one candidate that passed the static gates.  Its strict-'<' best /
            // second-best scan in visit order ends with the two smallest (distance, visit position) pairs = the two smallest
            // words, whatever order they are offered in.
            uint32_t bestW = W_INVALID, secondW = W_INVALID;
            auto consider = [&](uint32_t w) {
                const int k = (int)word_slot(w), dist = word_dist(w);
                const uint32_t st = state[k];
                if (MODE == MODE_PROJ || MODE == MODE_LAST) {
                    if ((st & ORBM_MP_PRESENT) && (st & ORBM_MP_OBS)) return;              // :113-115 / :1879-1881
                } else if (MODE != MODE_INIT) {
                    if (st & ORBM_MP_PRESENT) return;                                      // :2048 / :512
                }
                if (MODE == MODE_INIT) { if ((int)st <= dist) return; }                    // :608
                if (dist <= thr) atomicMin(&post[k], (uint32_t)tid);                       // this query may still claim k
                secondW = min(max(bestW, w), secondW);                                     // :124-137 etc.
                bestW = min(bestW, w);
            };
            // ---- evaluate ---------------------------------------------------------------------------
            if (unresolved) {
                if (cached) {
                    // wave-uniform trip count: the register array is indexed by a scalar (v_movrels), no unrolling
                    for (int i = 0; i < wave_ncand; i++) {
                        const uint32_t v = cand_at(i);
                        if (i < ncand) consider(v);
                    }
                } else {
                    auto body = [&](int k, int oct, float ur, int j) {
                        if (!static_gates<MODE>(cur, stereo_gate, oct, ur)) return;
                        const v4u lo = acc.desc[2 * k], hi = acc.desc[2 * k + 1];
                        const uint32_t cd[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                        consider((hamming256(qd, cd) << 17) | ((uint32_t)j << 4) | (uint32_t)oct);
                    };
                    if (MODE == MODE_SIM3) walk_window<true>(acc, x, y, r, minL, maxL, body);
                    else walk_window<false>(acc, x, y, r, minL, maxL, body);
                }
            }
            const int none = MODE == MODE_INIT ? INT_MAX : 256;                            // :594-595 / :100-103 etc.
            const bool has1 = word_valid(bestW), has2 = word_valid(secondW);
            const int bestDist = has1 ? word_dist(bestW) : none, bestDist2 = has2 ? word_dist(secondW) : none;
            const int bestIdx = has1 ? (int)word_slot(bestW) : -1, bestIdx2 = has2 ? (int)word_slot(secondW) : -1;
            const int bestLevel = has1 ? word_oct(bestW) : -1, bestLevel2 = has2 ? word_oct(secondW) : -1;
            // a query that can still be accepted also names its second-best free slot where a ratio test reads it: lower
            // queries' interest in a slot must be visible to whoever wants to take it (claimable slots are posted above)
            constexpr bool USES_SECOND = MODE == MODE_PROJ || MODE == MODE_INIT;
            const bool live = unresolved && bestIdx >= 0 && bestDist <= thr;
            // (as in fused_kernel: the second-best free slot matters only to a query whose ratio test does not pass on
            // distances now -- it can only get worse, so a test that passes now passes with any later second best)
            bool second_matters = false;
            if (USES_SECOND && live && bestIdx2 >= 0) {
                if (MODE == MODE_PROJ) second_matters = (float)bestDist > __fmul_rn(P.ratio, (float)bestDist2);
                else second_matters = !((float)bestDist < __fmul_rn((float)bestDist2, P.ratio));
            }
            if (second_matters) atomicMin(&post[bestIdx2], (uint32_t)tid);
            __syncthreads();
            // ---- contested? ---------------------------------------------------------------------------
            // The outcome is a function of the best and (where a ratio test reads it) the second-best free candidate, and
            // slots only ever become blocked.  If a lower-index unresolved query may still take one of those two, this
            // query is not settled, and ANY of its free candidates may end up in its top two: it posts on all of them so
            // that no later query takes one of them first.
            bool contested = false;
            if (live) {
                contested = post[bestIdx] < (uint32_t)tid;
                if (second_matters && post[bestIdx2] < (uint32_t)tid) contested = true;
            }
            if (wg_any(contested)) {
                if (contested) {
                    if (cached) {
                        for (int i = 0; i < wave_ncand; i++) {
                            const uint32_t v = cand_at(i);
                            if (i < ncand) {
                                const int k = (int)word_slot(v);
                                if (!is_blocked(state[k], word_dist(v))) atomicMin(&post[k], (uint32_t)tid);
                            }
                        }
                    } else {
                        auto body = [&](int k, int oct, float ur, int j) {
                            if (!static_gates<MODE>(cur, stereo_gate, oct, ur)) return;
                            const int slot = j;
                            if (MODE == MODE_INIT) {
                                const v4u lo = acc.desc[2 * k], hi = acc.desc[2 * k + 1];
                                const uint32_t cd[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                                if (is_blocked(state[slot], (int)hamming256(qd, cd))) return;
                            } else if (is_blocked(state[slot], 0)) return;
                            atomicMin(&post[slot], (uint32_t)tid);
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
                        const int bestKp = (int)F.cell_idx[bestIdx];                      // the slot's keypoint
                        if (MODE == MODE_INIT) {
                            const int old = owner[bestIdx];
                            if (old >= 0) { out[old] = -1; my_matches--; }                 // :629-633
                            out[qi] = bestKp; owner[bestIdx] = qi; state[bestIdx] = (uint32_t)bestDist;  // :636-638
                        } else {
                            uint32_t ns = ORBM_MP_PRESENT;
                            if ((MODE != MODE_PROJ && MODE != MODE_LAST) || (qflags & ORBM_Q_OBS)) ns |= ORBM_MP_OBS;
                            state[bestIdx] = ns;
                            owner[bestIdx] = qi;                                           // :160 / :1906 / :2066 / :535
                        }
                        my_matches++;
                        if (use_hist) {
                            const int bin = rot_bin(qangle, F.angle[bestKp]);              // :644-651 / :1911-1918 / :2079-2090
                            atomicAdd(&hist[bin], 1);
                            if (MODE != MODE_INIT) atomicOr(&binmask[bestIdx], 1u << bin);
                        }
                    }
                }
            }
            // the other post buffer is idle in this phase (decide only reads `post`): clear it for the next round
            for (int i = tid; i < n_slots; i += nthr) post_next[i] = NOPOST;
            if (PRE && round < 40) WIN_STAMP(21 + round);
            if (!wg_any(unresolved)) break;
            round++;
        }
        // leave both post buffers clean for the next chunk
        {
            uint32_t *post = (round & 1) ? post1 : post0;
            for (int i = tid; i < n_slots; i += nthr) post[i] = NOPOST;
        }
        __syncthreads();

    };

    if (!PRE) {
        // single-kernel form: queries in index order, the NEXT chunk's query is loaded while the current one is matched
        Query nextq = load_query(tid, false);
        for (int base_q = 0; base_q < Q; base_q += nthr) {
            const int qi = base_q + tid;
            const Query cur = nextq;
            nextq = load_query(qi + nthr, false);
            uint32_t cc[WIN_CACHE];
#pragma unroll
            for (int i = 0; i < WIN_CACHE; i++) cc[i] = 0;
            process_chunk(qi, cur, 0, true, cc);
        }
    } else {
        // split form.  collect_kernel left every query's smallest candidate words (head: four, 16 bytes).  A query none
        // of whose claimable candidates (distance <= threshold) is still free can never be accepted, and blocked slots stay
        // blocked, so the queries are first scanned SCAN_U x blockDim at a time against the current slot state (a few
        // instructions each, loads independent) and only the live ones, compacted in index order, go through the
        // rounds.  Typical inputs leave a few percent of the queries live.
        uint16_t *qlist = reinterpret_cast<uint16_t *>(hist + 64);          // SCAN_U x blockDim
        int *cnts = hist + 64 + SCAN_U * nthr / 2;                          // SCAN_U x waves (+1)
        int *pre = cnts + SCAN_U * 16 + 1;
        const int wave = tid >> 6, lane = tid & 63, NW = nthr >> 6;
        for (int sb = 0; sb < Q; sb += SCAN_U * nthr) {
            v4u w[SCAN_U];
#pragma unroll
            for (int u = 0; u < SCAN_U; u++) {
                const int qi = sb + u * nthr + tid;
                w[u] = qi < Q ? heads[qi] : v4u{W_INVALID, W_INVALID, W_INVALID, W_INVALID};
            }
            uint32_t lm = 0, ranks = 0, rankhi = 0;
#pragma unroll
            for (int u = 0; u < SCAN_U; u++) {
                const uint32_t hw[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
                // words ascend: if the fourth is claimable and a tail exists, the tail may hold claimable ones too
                const int tc = word_tail(hw[2], hw[3]);
                bool live = tc == TAIL_UNKNOWN || (tc > 0 && word_dist(hw[3]) <= thr);
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (word_valid(hw[j]) && word_dist(hw[j]) <= thr && !is_blocked(state[word_slot(hw[j])], word_dist(hw[j]))) live = true;
                const uint64_t m = __ballot(live);
                if (lane == 0) cnts[u * NW + wave] = __popcll(m);
                const uint32_t rk = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (u < 4) ranks |= rk << (8 * u); else rankhi |= rk << (8 * (u - 4));
                if (live) lm |= 1u << u;
            }
            __syncthreads();
            if (tid < 64) {                                                  // exclusive prefix over the SCAN_U x NW (<= 128) counts, one wave
                const int a = 2 * tid < SCAN_U * NW ? cnts[2 * tid] : 0, b = 2 * tid + 1 < SCAN_U * NW ? cnts[2 * tid + 1] : 0;
                int incl = a + b;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int up = __shfl_up(incl, d, 64);
                    if (tid >= d) incl += up;
                }
                const int excl = incl - (a + b);
                if (2 * tid < SCAN_U * NW) pre[2 * tid] = excl;
                if (2 * tid + 1 < SCAN_U * NW) pre[2 * tid + 1] = excl + a;
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
            WIN_STAMP(18);
#ifdef ORBM_DIAG
            if (blockIdx.x == 0 && tid == 0) d_win_stamps[70] = (unsigned long long)nlive;
#endif
            for (int c0 = 0; c0 < nlive; c0 += nthr) {
                const int li = c0 + tid;
                const int qi = li < nlive ? sb + (int)qlist[li] : Q;
                uint32_t cc[WIN_CACHE];
#pragma unroll
                for (int i = 0; i < WIN_CACHE; i++) cc[i] = W_INVALID;
                bool complete = true;
                int nc = 0;
                if (qi < Q) {
                    const v4u h4 = heads[qi];
                    cc[0] = h4.x; cc[1] = h4.y; cc[2] = h4.z; cc[3] = h4.w;
                    const int tc = word_tail(h4.z, h4.w);
                    if (word_valid(h4.w)) { cc[2] = h4.z & ~W_FLAGS; cc[3] = h4.w & ~W_FLAGS; }
                    complete = tc != TAIL_UNKNOWN;
                    if (complete && tc > 0) {
                        const gptr<const v4u> tq = (gptr<const v4u>)(tails + (size_t)qi * tcap);   // tcap is a multiple of 4
#pragma unroll
                        for (int g = 0; g < (WIN_CACHE - 4) / 4; g++)
                            if (tc > 4 * g) {
                                const v4u t4 = tq[g];
                                cc[4 + 4 * g] = t4.x;
                                cc[5 + 4 * g] = tc > 4 * g + 1 ? t4.y : W_INVALID;
                                cc[6 + 4 * g] = tc > 4 * g + 2 ? t4.z : W_INVALID;
                                cc[7 + 4 * g] = tc > 4 * g + 3 ? t4.w : W_INVALID;
                            }
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++) nc += word_valid(cc[i]) ? 1 : 0;
                    nc = complete ? nc + tc : tcap + 5;                     // more candidates than words: the rounds walk the window
                }
                const Query cur = load_query(qi, !complete);
                process_chunk(qi, cur, nc, complete, cc);
            }
        }
    }

    if (my_matches) atomicAdd(nmatch, my_matches);
    __syncthreads();
    if (PRE) WIN_STAMP(62);
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
            for (int i = tid; i < n_slots; i += nthr)
                if (binmask[i] & ~keepmask) owner[i] = -2;                                 // slot set to NULL (:1942 / :2115)
        }
        __syncthreads();
    }
    if (MODE == MODE_INIT) {
        for (int i = tid; i < Q; i += nthr) {                                             // :684-686
            const int m = out[i];
            if (m >= 0) { P.prev_out[2 * i] = F.x[m]; P.prev_out[2 * i + 1] = F.y[m]; }
        }
    } else {
        // keypoints outside the grid (no stream position) are never candidates: -1
        if (n_slots != n) {
            for (int i = tid; i < n; i += nthr) out[i] = -1;
            __syncthreads();
        }
        for (int i = tid; i < n_slots; i += nthr) out[F.cell_idx[i]] = owner[i];
    }
    if (tid == 0) P.counts[inst] = *nmatch;
    if (PRE) WIN_STAMP(63);
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
        auto body = [&](int k, int oct, float kur, int) {
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
                  const v4u *head, const uint32_t *tail, int tcap = 0, const int32_t *redo_flags = nullptr) {
    const size_t lds = ((size_t)(LDS_FRAME ? 3076 + 12 * (size_t)max_n : 0) + 5 * (size_t)max_n + 64 +
                        (PRE ? SCAN_U * threads / 2 + 2 * (SCAN_U * 16 + 1) : 0)) * 4;
    if (lds > 160 * 1024)
        return orbm_fail(ctx, ORBM_ELIMIT, "windowed search: frames of %d keypoints need %zu bytes of LDS slot state (limit 163840)", max_n, lds);
    HIPCHK(ctx, (orbm_allow_dynamic_lds<&window_kernel<MODE, LDS_FRAME, PRE>>(ctx, lds)));
    hipLaunchKernelGGL((window_kernel<MODE, LDS_FRAME, PRE>), dim3(n_inst), dim3(threads), lds, ctx->stream, P,
                       waves_per_inst, head, tail, tcap, redo_flags);
    HIPCHK(ctx, hipGetLastError());
    return ORBM_OK;
}

// The index collect_kernel builds in LDS (see there): structures, coarse-column shifts, bucket bases.  `pairs`: one
// structure per octave pair; rmax_px[p] = the largest window half-size expected at level p (efficiency only: a window
// of any size is walked correctly, it just spans more coarse columns).
static CoarseCfg coarse_cfg(bool pairs, const float *rmax_px, int max_n) {
    CoarseCfg c;
    memset(&c, 0, sizeof c);
    c.nstruct = pairs ? 8 : 1;
    int b = 0;
    for (int s2 = 0; s2 < c.nstruct; s2++) {
        // fine columns spanned: 2r/10 + 2..3; aim at no more than ~4 coarse columns per window
        const float need = (2.0f * rmax_px[s2] / 10.0f + 2.0f) / 3.0f;
        int k = 0;
        while ((1 << k) < need && k < 3) k++;
        c.kshift[s2] = k;
        c.base[s2] = b;
        b += (ORBM_GRID_COLS >> k) * ORBM_GRID_ROWS;
    }
    for (int s2 = c.nstruct; s2 <= 8; s2++) c.base[s2] = b;
    c.rec_cap = (pairs ? 2 : 1) * max_n + 4;
    c.n_cap = max_n;
    return c;
}

// dynamic LDS of collect_kernel: bucket offsets (u16), one record per stream position, the buckets' runs (u16 stream
// positions), parking (u16), cost classes, lane assignment (u16) and, when it fits (frames up to ~2100 keypoints), the
// frame's descriptors in stream order
static size_t collect_lds_bytes(const CoarseCfg &c, bool desc_lds) {
    const int total = c.base[c.nstruct];
    const size_t offn = (size_t)((total + 6) & ~1);
    return (((offn / 2 + 3) & ~(size_t)3)) * 4 + (size_t)((c.n_cap + 3) & ~3) * 16 + (size_t)((c.rec_cap + 7) & ~7) * 2 +
           (size_t)PARK * COLLECT_THREADS * 2 + COST_CLASSES * 4 + 2 * COLLECT_THREADS * 2 + 64 + (desc_lds ? (size_t)c.n_cap * 32 + 16 : 0);
}

// tuning / test knob: 0 = split form (collect_kernel + claims), 1 = single kernel with the frame staged in LDS,
// 2 = single kernel reading the frame from global memory
// 3 = the fused form (one launch per batch) wherever it fits, 4 = never the fused form; 0 = by batch shape:
int g_win_variant = 0;
// (measured, scripts/track_perf.py + proj_fused_perf.py: one instance alone is faster split -- its search spreads over the
// chip, 0.076 against 0.112 ms for 1500 queries -- except SearchForInitialization, whose long claim chains dominate; a batch
// is faster fused from a few dozen instances on: 256 x 50k queries 0.24 against 0.88 ms)
int g_win_fused_min_inst = 48;     // fused from this many instances per call ...
int g_win_fused_max_q = 0;         // ... or when no query set of the call is larger than this (0: never by size)
int g_win_coop = 1;                // split form: calls with few queries spread every query's candidates over a wave's lanes (test knob)
int g_win_cost_exact = 1;   // cost classes of collect_kernel: 1 = the walk's exact trip count, 0 = window area (A/B knob)

template <int MODE>
int launch_mode(orbm_ctx *ctx, const WinParams &P, int n_inst, int max_n, int max_q, int max_levels) {
    // window half-sizes this launch will see, per level (nominal 1.2 pyramid; only steers the coarse-column width)
    float rmax[8];
    float sf = 1.0f;
    for (int l = 0; l < 8; l++, sf *= 1.2f) {
        if (MODE == MODE_PROJ) rmax[l] = 4.0f * (P.th != 1.0f ? P.th : 1.0f) * sf;
        else if (MODE == MODE_INIT) rmax[l] = (float)P.window;
        else rmax[l] = 10.0f * sf;
    }
    // a6, a5 and the Sim3 overload ask for one octave pair per query; the other two overloads for wider or open ranges
    constexpr bool PAIRS = MODE == MODE_PROJ || MODE == MODE_INIT || MODE == MODE_SIM3;
    if (!PAIRS) rmax[0] = 20.0f;
    const CoarseCfg cfg = coarse_cfg(PAIRS, rmax, max_n);
    // frames too large for collect_kernel's LDS index (or with more than 8 pyramid levels) use the single-kernel form
    const bool index_ok = max_n <= 8191 && !(PAIRS && max_levels > 8);
    const bool single = g_win_variant == 1 || g_win_variant == 2 || collect_lds_bytes(cfg, false) > 160 * 1024 || !index_ok;
    // the fused form (one launch, one workgroup per instance): batches that give the chip an instance per CU or so, and calls
    // with few queries (a tracking frame: one launch instead of two, no scratch round trip); a single large query set alone is
    // spread over the chip by the split form instead
    constexpr bool BINMASK = MODE == MODE_LAST || MODE == MODE_KF || MODE == MODE_INIT;
    constexpr size_t FUSED_LDS_MAX = 160 * 1024 - 1024;       // dynamic part: the kernel's static __shared__ arrays take the rest
    // (keeping the frame's descriptors in LDS as well was measured no faster, for a6 / a11 batches -- the search of the first
    // chunks is bound by instruction issue, not by the candidate gathers -- and for a5 alone, whose few hundred valid queries
    // walk some thirty candidates each, one after the other: 127 against 129 us per call; the 32 B per keypoint stay free)
    const size_t fused_lds = fused_lds_bytes(cfg, BINMASK) + (MODE == MODE_INIT ? fused_init_extra_bytes(cfg) : 0);
    const bool fused_fits = index_ok && fused_lds <= FUSED_LDS_MAX && max_q > 0;
    if (fused_fits && g_win_variant != 1 && g_win_variant != 2 && g_win_variant != 4 &&
        (g_win_variant == 3 || MODE == MODE_INIT || n_inst >= g_win_fused_min_inst || max_q <= g_win_fused_max_q)) {
        constexpr size_t WIN_SCRATCH_CAP = (size_t)1 << 30;
        int tcap = 60;
        while (tcap > 12 && (size_t)n_inst * 2 * FUSED_CHUNK * tcap * 4 > WIN_SCRATCH_CAP) tcap -= 16;
        const size_t inst_bytes = (size_t)2 * FUSED_CHUNK * tcap * 4;
        const int run_max = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_inst, WIN_SCRATCH_CAP / inst_bytes));
        const size_t o_flags = ((size_t)run_max * inst_bytes + 255) & ~(size_t)255, o_trace = o_flags + (((size_t)run_max * 4 + 255) & ~(size_t)255);
        int rc = orbm_ws2_reserve(ctx, o_trace + 16384);
        if (rc) return rc;
        // the pass that searches flagged instances again (the general single-kernel form reading global memory) must be launchable
        if (((size_t)5 * max_n + 64) * 4 > 160 * 1024) return orbm_fail(ctx, ORBM_ELIMIT, "windowed search: frames of %d keypoints are not supported", max_n);
        HIPCHK(ctx, (orbm_allow_dynamic_lds<&fused_kernel<MODE>>(ctx, fused_lds)));
        for (int i0 = 0; i0 < n_inst; i0 += run_max) {
            const int ni = std::min(run_max, n_inst - i0);
            WinParams R = P;
            R.frame_ids += i0; R.qset_ids += i0; R.out += (size_t)i0 * P.stride; R.counts += i0;
            if (R.slot_flags) R.slot_flags += (size_t)i0 * P.stride;
            unsigned long long *trace = nullptr;
#ifdef ORBM_DIAG
            static const bool want_trace = getenv("ORBM_WIN_TRACE") != nullptr;
            if (want_trace) {                                  // timestamps of instance 0, after the tail scratch
                trace = (unsigned long long *)((char *)ctx->ws2 + o_trace);
                HIPCHK(ctx, hipMemsetAsync(trace, 0, 16384, ctx->stream));
            }
#endif
            int32_t *redo = (int32_t *)((char *)ctx->ws2 + o_flags);
            hipLaunchKernelGGL((fused_kernel<MODE>), dim3((unsigned)ni), dim3(FUSED_THREADS), fused_lds, ctx->stream, R, cfg, (uint32_t *)ctx->ws2, tcap, redo, trace);
            HIPCHK(ctx, hipGetLastError());
            // instances with a query that has more candidates than the fused form holds (tcap + 8: very wide windows over crowded
            // keypoints) are searched again by the general form; everywhere else its workgroups return at once
            rc = launch_claims<MODE, false, false>(ctx, R, ni, max_n, MODE == MODE_INIT ? 256 : 512, 0, nullptr, nullptr, 0, redo);
            if (rc) return rc;
#ifdef ORBM_DIAG
            if (want_trace) {
                static unsigned long long h[2048];
                HIPCHK(ctx, hipMemcpyAsync(h, trace, 16384, hipMemcpyDeviceToHost, ctx->stream));
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
                fprintf(stderr, "[fused trace] index %.2f us, total %.2f us; per chunk (search+handoff us | claims us | live):", (h[1] - h[0]) / 100.0, (h[599] - h[0]) / 100.0);
                unsigned long long prev = h[1];
                for (int c = 0; c < 250 && h[2 + 2 * c]; c++) {
                    fprintf(stderr, " [%d: %.2f (search %.2f) | %.2f | %llu]", c, (h[2 + 2 * c] - prev) / 100.0, c < 190 ? (h[800 + c] - prev) / 100.0 : 0.0,
                            (h[3 + 2 * c] - h[2 + 2 * c]) / 100.0, h[600 + c]);
                    prev = h[2 + 2 * c];
                }
                fprintf(stderr, "\n[fused trace] chunk 0 batches (load us | rounds us | rounds | widest list):");
                for (int b = 0; b < 15 && h[1000 + 4 * b]; b++)
                    fprintf(stderr, " [%.2f | %.2f | %llu | %llu]", (h[1001 + 4 * b] - h[1000 + 4 * b]) / 100.0, (h[1002 + 4 * b] - h[1001 + 4 * b]) / 100.0,
                            h[1003 + 4 * b] & 0xFFFFFFFFull, h[1003 + 4 * b] >> 32);
                fprintf(stderr, "\n");
            }
#endif
        }
        return ORBM_OK;
    }
    if (single) {
        if (max_n <= WIN_LDS_FRAME_MAX_N && g_win_variant != 2)
            return launch_claims<MODE, true, false>(ctx, P, n_inst, max_n, MODE == MODE_INIT ? 256 : 1024, 0, nullptr, nullptr);
        return launch_claims<MODE, false, false>(ctx, P, n_inst, max_n, MODE == MODE_INIT ? 256 : 512, 0, nullptr, nullptr);
    }
    {
        const int waves_per_inst = (max_q + 63) / 64;
        if (waves_per_inst == 0) return launch_claims<MODE, false, false>(ctx, P, n_inst, max_n, 256, 0, nullptr, nullptr);
        // per query: 16 bytes of head (always written) + room for tcap tail words (written for queries with more than 4
        // candidates).  The scratch (ws2, grow-only for the context's lifetime) is capped at WIN_SCRATCH_CAP: first the tail
        // shrinks (60 -> 28 words; queries with more candidates than that are still exact, the claims kernel walks their window
        // again), then the batch is cut into runs of instances that go through the two kernels one after the other.
        constexpr size_t WIN_SCRATCH_CAP = (size_t)1 << 30;
        const size_t q_inst = (size_t)waves_per_inst * 64;
        int tcap = 60;
        while (tcap > 28 && (size_t)n_inst * q_inst * (16 + 4 * (size_t)tcap) > WIN_SCRATCH_CAP) tcap -= 16;
        const size_t inst_bytes = q_inst * (16 + 4 * (size_t)tcap);
        const int run_max = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_inst, WIN_SCRATCH_CAP / inst_bytes));
        int rc = orbm_ws2_reserve(ctx, (size_t)run_max * inst_bytes + 256);
        if (rc) return rc;
        // enough workgroups to fill the chip, each with enough rows to pay for building the index
        constexpr int WAVES = COLLECT_THREADS / 64;
        // one workgroup per CU is enough (the LDS index allows no second one) and every workgroup builds the index once
        static const int wg_target_env = [] { const char *e = getenv("ORBM_WIN_WGS"); return e ? atoi(e) : 0; }();
        const int wg_target = wg_target_env > 0 ? wg_target_env : std::max(64, ctx->prop.multiProcessorCount);
        const bool desc_lds = collect_lds_bytes(cfg, true) <= 160 * 1024 && !(g_win_cost_exact & 32);
        const size_t lds = collect_lds_bytes(cfg, desc_lds);
        HIPCHK(ctx, (orbm_allow_dynamic_lds<&collect_kernel<MODE>>(ctx, lds)));
        const int threads = max_q >= 1024 ? 1024 : (max_q > 256 ? 512 : 256);
        for (int i0 = 0; i0 < n_inst; i0 += run_max) {
            const int ni = std::min(run_max, n_inst - i0);
            WinParams R = P;                                   // this run's instances: every per-instance array moved up by i0
            R.frame_ids += i0; R.qset_ids += i0; R.out += (size_t)i0 * P.stride; R.counts += i0;
            if (R.slot_flags) R.slot_flags += (size_t)i0 * P.stride;
            v4u *head = (v4u *)ctx->ws2;
            uint32_t *tail = (uint32_t *)(head + (size_t)ni * q_inst);
            int wgs_per_inst = std::max(1, std::min((wg_target + ni - 1) / ni, (waves_per_inst + 2 * WAVES - 1) / (2 * WAVES)));
            // few instances: one row of 64 queries per wave instead of two, so that an instance spreads over twice the workgroups
            const int groups = (long long)ni * wgs_per_inst < wg_target ? 1 : 2;
            if (groups == 1) wgs_per_inst = std::max(1, std::min((wg_target + ni - 1) / ni, (waves_per_inst + WAVES - 1) / WAVES));
            // still fewer workgroups than CUs (one large instance): narrower rows, so that the instance spreads over the chip
            int qt = COLLECT_THREADS;
            if (groups == 1 && (long long)ni * wgs_per_inst < wg_target) {
                const long long per_wg = ((long long)waves_per_inst * 64 * ni + wg_target - 1) / wg_target;
                qt = (int)std::min<long long>(COLLECT_THREADS, std::max<long long>(64, (per_wg + 63) / 64 * 64));
                wgs_per_inst = std::max(1, (waves_per_inst * 64 + qt - 1) / qt);
            }
            // a call with few queries in all (a tracking frame): one or two queries per WAVE, a query's candidates over the lanes
            int coop = 0;
            const long long total_q = (long long)ni * waves_per_inst * 64;
            if (g_win_coop && MODE != MODE_INIT && total_q <= 2LL * WAVES * wg_target) {
                coop = total_q <= (long long)WAVES * wg_target ? WAVES : 2 * WAVES;
                wgs_per_inst = (waves_per_inst * 64 + coop - 1) / coop;
            }
            hipLaunchKernelGGL((collect_kernel<MODE>), dim3((unsigned)(ni * wgs_per_inst)), dim3(COLLECT_THREADS), lds, ctx->stream, R,
                               waves_per_inst, wgs_per_inst, head, tail, tcap, cfg, groups, qt, g_win_cost_exact, desc_lds ? 1 : 0, coop);
            HIPCHK(ctx, hipGetLastError());
            rc = launch_claims<MODE, false, true>(ctx, R, ni, max_n, threads, waves_per_inst, head, tail, tcap);
            if (rc) return rc;
        }
        return ORBM_OK;
    }
    return ORBM_OK;
}

int launch_window(orbm_ctx *ctx, int mode, const WinParams &P, int n_inst, int max_n, int max_q, int max_levels) {
    switch (mode) {
        case MODE_LAST: return launch_mode<MODE_LAST>(ctx, P, n_inst, max_n, max_q, max_levels);
        case MODE_KF: return launch_mode<MODE_KF>(ctx, P, n_inst, max_n, max_q, max_levels);
        case MODE_SIM3: return launch_mode<MODE_SIM3>(ctx, P, n_inst, max_n, max_q, max_levels);
        case MODE_INIT: return launch_mode<MODE_INIT>(ctx, P, n_inst, max_n, max_q, max_levels);
        default: return launch_mode<MODE_PROJ>(ctx, P, n_inst, max_n, max_q, max_levels);
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
    int max_n = 0, max_q = 0, max_levels = 0;
    if (!io_on_device) {
        int rc = orbm_check_frames(ctx, n_inst, frame, "frame");
        if (!rc) rc = orbm_check_qsets(ctx, n_inst, qset, mode == MODE_PROJ ? 0 : 1, "qset");
        if (rc) return rc;
        for (int k = 0; k < n_inst; k++) {
            max_n = std::max(max_n, ctx->h_frames[frame[k]].n);
            max_levels = std::max(max_levels, ctx->h_frames[frame[k]].n_levels);
            max_q = std::max(max_q, ctx->h_qsets[qset[k]].n);
        }
        if (max_n > stride) return orbm_fail(ctx, ORBM_EINVAL, "%s: stride %d < largest frame %d", who, stride, max_n);
    } else {
        max_n = stride;  // the caller's stride bounds every frame of the batch
        // the ids live on the device: bound the query count by the largest live query set of the context
        for (size_t k = 0; k < ctx->h_qsets.size(); k++)
            if (ctx->qset_arena[k] >= 0) max_q = std::max(max_q, ctx->h_qsets[k].n);
        // ... the keypoints (LDS sizing) and the pyramid depth by the live frames of the context
        int live_n = 0;
        for (size_t k = 0; k < ctx->h_frames.size(); k++)
            if (ctx->frame_arena[k] >= 0) { live_n = std::max(live_n, ctx->h_frames[k].n); max_levels = std::max(max_levels, ctx->h_frames[k].n_levels); }
        max_n = std::min(max_n, live_n);
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
    // rows beyond a frame's n are part of what host mode copies back: -1 there (the kernels write a frame's own n rows)
    bool ragged = false;
    if (!io_on_device) for (int k = 0; k < n_inst && !ragged; k++) ragged = ctx->h_frames[frame[k]].n != stride;
    if (ragged && out_elems) HIPCHK(ctx, hipMemsetAsync(down, 0xFF, out_elems * 4, ctx->stream));
    WinParams P;
    memset(&P, 0, sizeof P);
    P.frames = ctx->d_frames; P.qsets = ctx->d_qsets;
    P.frame_ids = (const int32_t *)dfr; P.qset_ids = (const int32_t *)dqs;
    P.slot_flags = (const uint8_t *)dsf;
    P.out = (int32_t *)down; P.counts = (int32_t *)dc;
    P.th = th; P.ratio = ratio; P.dist_th = dist_th; P.check_ori = check_ori; P.stride = stride;
    rc = launch_window(ctx, mode, P, n_inst, std::max(max_n, 1), max_q, max_levels);
    if (rc) return rc;
    return io.finish();
}

}  // namespace

// tuning / test knob (not part of the public ABI): see g_win_variant
#ifdef ORBM_DIAG
extern "C" void orbm__window_stamps(unsigned long long *out, int clear) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(d_win_stamps), sizeof(unsigned long long) * 128);
    if (clear) { static unsigned long long z[128]; (void)hipMemcpyToSymbol(HIP_SYMBOL(d_win_stamps), z, sizeof(z)); }
}
#endif
extern "C" void orbm__window_variant(int v) { g_win_variant = v; }
extern "C" void orbm__window_fused_rule(int min_inst, int max_q) { g_win_fused_min_inst = min_inst; g_win_fused_max_q = max_q; }
extern "C" void orbm__window_coop(int on) { g_win_coop = on; }
#ifdef ORBM_DIAG
// phase-timing diagnostics of collect_kernel (bits 1-3 switch phases off: wrong results); only in `make diag` builds
extern "C" void orbm__window_cost_exact(int v) { g_win_cost_exact = v; }
#endif

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

// ---- one call = one search: the query set travels WITH the call.  What a tracking thread issues per frame
// (Tracking.cc:1152, 1549-1560, 1865) is one search of one fresh query set into one resident frame; through
// orbm_*_queries_upload + orbm_search_*_batch + orbm_queries_free that is three calls, three stream synchronisations, an
// arena and two table updates.  Here the query arrays, a one-entry query-set table and the id lists are packed into the
// context's pinned staging block and go to the device as ONE copy, the kernels read the set where it landed, and outputs
// come back as one copy: a single synchronisation, nothing allocated, nothing to free.  Results are those of the batch
// entry points with n_inst = 1.
static int search_once(orbm_ctx *ctx, int mode, int32_t frame, const orbm_proj_queries *pq, const orbm_win_queries *wq, float th, float ratio,
                       int dist_th, int check_ori, const uint8_t *slot_flags, int32_t *owner, int32_t *count, const char *who) {
    if (!ctx) return ORBM_EINVAL;
    if ((!pq && !wq) || !owner || !count) return orbm_fail(ctx, ORBM_EINVAL, "%s: null argument", who);
    int rc = orbm_check_frames(ctx, 1, &frame, "frame");
    if (rc) return rc;
    const int nq = pq ? pq->n : wq->n;
    const int nF = ctx->h_frames[frame].n;
    if (nq < 0) return orbm_fail(ctx, ORBM_EINVAL, "%s: n < 0", who);
    if (nF > WIN_MAX_N) return orbm_fail(ctx, ORBM_ELIMIT, "%s: frames of more than %d keypoints are not supported by the windowed kernels (got %d)", who, WIN_MAX_N, nF);
    if (pq && nq > 0 && (!pq->desc || !pq->proj_x || !pq->proj_y || !pq->proj_xr || !pq->level || !pq->view_cos))
        return orbm_fail(ctx, ORBM_EINVAL, "%s: null query array", who);
    if (wq && nq > 0 && (!wq->desc || !wq->u || !wq->v || !wq->radius || !wq->min_level || !wq->max_level))
        return orbm_fail(ctx, ORBM_EINVAL, "%s: null query array", who);
    *count = 0;
    if (nF == 0) return ORBM_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)nq;
    rc = orbm_ws_reserve(ctx, n * 64 + 16 * 256 + (size_t)nF * 5 + 8192);
    if (rc) return rc;
    IoStage io(ctx, 0);
    QSetDev qs;
    memset(&qs, 0, sizeof qs);
    qs.n = nq; qs.kind = pq ? 0 : 1;
    const void *d;
    std::vector<uint32_t> packed;
    std::vector<uint8_t> fl;
    if (pq) {
        for (size_t i = 0; i < n; i++)
            if (pq->level[i] < 0 || pq->level[i] >= ORBM_MAX_LEVELS) return orbm_fail(ctx, ORBM_EINVAL, "%s: level[%zu] out of range", who, i);
        packed.resize(4 * n + 4);
        for (size_t i = 0; i < n; i++) {
            memcpy(&packed[4 * i], &pq->proj_x[i], 4); memcpy(&packed[4 * i + 1], &pq->proj_y[i], 4); memcpy(&packed[4 * i + 2], &pq->proj_xr[i], 4);
            const uint32_t f = pq->flags ? pq->flags[i] : (uint32_t)(ORBM_Q_VALID | ORBM_Q_OBS);
            packed[4 * i + 3] = (uint32_t)pq->level[i] | (f << 8) | (pq->view_cos[i] >= 0.998f ? 1u << 16 : 0u);   // ORBmatcher.cc:168-174
        }
        if ((rc = io.in(n ? pq->desc : (const uint8_t *)packed.data(), std::max<size_t>(n * 32, 16), &d))) return rc;
        qs.desc = (const uint32_t *)d;
        if ((rc = io.in(packed.data(), std::max<size_t>(n * 16, 16), &d))) return rc;
        qs.pk = (const uint32_t *)d;
    } else {
        static const uint8_t zero16[16] = {0};
        auto put = [&](const void *p, size_t bytes, const void **out) { return io.in(n ? p : (const void *)zero16, std::max<size_t>(bytes, 16), out); };
        if ((rc = put(wq->desc, n * 32, &d))) return rc; qs.desc = (const uint32_t *)d;
        if ((rc = put(wq->u, n * 4, &d))) return rc; qs.f0 = (const float *)d;
        if ((rc = put(wq->v, n * 4, &d))) return rc; qs.f1 = (const float *)d;
        if ((rc = put(wq->radius, n * 4, &d))) return rc; qs.f2 = (const float *)d;
        if (wq->ur) { if ((rc = put(wq->ur, n * 4, &d))) return rc; qs.f3 = (const float *)d; }
        if (wq->angle) { if ((rc = put(wq->angle, n * 4, &d))) return rc; qs.angle = (const float *)d; }
        if ((rc = put(wq->min_level, n * 4, &d))) return rc; qs.l0 = (const int32_t *)d;
        if ((rc = put(wq->max_level, n * 4, &d))) return rc; qs.l1 = (const int32_t *)d;
        if (wq->flags) { if ((rc = put(wq->flags, n, &d))) return rc; }
        else { fl.assign(std::max<size_t>(n, 16), ORBM_Q_VALID | ORBM_Q_OBS); if ((rc = io.in(fl.data(), fl.size(), &d))) return rc; }
        qs.flags = (const uint8_t *)d;
    }
    const void *dtab, *dids, *dsf = nullptr; void *down, *dc;
    const int32_t ids[4] = {frame, 0, 0, 0};              // frame_ids[0] = frame, qset_ids[0] = 0 (the one-entry table below)
    if ((rc = io.in(&qs, sizeof qs, &dtab)) || (rc = io.in(ids, sizeof ids, &dids)) || (rc = io.in(slot_flags, (size_t)nF, &dsf)) ||
        (rc = io.out(owner, (size_t)nF * 4, &down)) || (rc = io.out(count, 4, &dc)))
        return rc;
    WinParams P;
    memset(&P, 0, sizeof P);
    P.frames = ctx->d_frames; P.qsets = (const QSetDev *)dtab;
    P.frame_ids = (const int32_t *)dids; P.qset_ids = (const int32_t *)dids + 1;
    P.slot_flags = (const uint8_t *)dsf;
    P.out = (int32_t *)down; P.counts = (int32_t *)dc;
    P.th = th; P.ratio = ratio; P.dist_th = dist_th; P.check_ori = check_ori; P.stride = nF;
    rc = launch_window(ctx, mode, P, 1, std::max(nF, 1), nq, ctx->h_frames[frame].n_levels);
    if (rc) return rc;
    return io.finish();
}

extern "C" int orbm_search_proj_points_once(orbm_ctx *ctx, int32_t frame, const orbm_proj_queries *queries, float th, float ratio,
                                            const uint8_t *slot_flags, int32_t *owner, int32_t *count) {
    return search_once(ctx, MODE_PROJ, frame, queries, nullptr, th, ratio, ORBM_TH_HIGH, 0, slot_flags, owner, count, "search_proj_points_once");
}

extern "C" int orbm_search_window_once(orbm_ctx *ctx, int mode, int32_t frame, const orbm_win_queries *queries, int dist_th, int check_ori,
                                       const uint8_t *slot_flags, int32_t *owner, int32_t *count) {
    if (ctx && (mode < 0 || mode > 2)) return orbm_fail(ctx, ORBM_EINVAL, "search_window_once: mode must be 0, 1 or 2");
    return search_once(ctx, mode, frame, nullptr, queries, 1.0f, 1.0f, dist_th, check_ori, slot_flags, owner, count, "search_window_once");
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
    // one pinned copy in, one out (as every other entry point; separate copies from and to pageable memory were a third of
    // this call).  vbPrevMatched is in / out: it goes in whole, the kernel writes the matched entries to a second array and
    // the host puts those over the caller's.
    rc = orbm_ws_reserve(ctx, (size_t)n1 * 20 + 4096);
    if (rc) return rc;
    IoStage io(ctx, 0);
    const int32_t ids[2] = {frame2, frame1};
    std::vector<float> prev_new((size_t)n1 * 2);
    int32_t count = 0;
    const void *d_ids, *d_prev; void *d_count, *d_prev_out, *d_m12;
    if ((rc = io.in(ids, 8, &d_ids)) || (rc = io.in(prev_xy, (size_t)n1 * 8, &d_prev)) || (rc = io.out(&count, 4, &d_count)) ||
        (rc = io.out(prev_new.data(), (size_t)n1 * 8, &d_prev_out)) || (rc = io.out(matches12, (size_t)n1 * 4, &d_m12)))
        return rc;
    WinParams P;
    memset(&P, 0, sizeof P);
    P.frames = ctx->d_frames; P.qsets = ctx->d_qsets;
    P.frame_ids = (const int32_t *)d_ids; P.qset_ids = (const int32_t *)d_ids + 1;
    P.prev_xy = (const float *)d_prev; P.prev_out = (float *)d_prev_out; P.out = (int32_t *)d_m12; P.counts = (int32_t *)d_count;
    P.th = 1.0f; P.ratio = ratio; P.dist_th = ORBM_TH_LOW; P.check_ori = check_ori; P.stride = n1; P.window = window;
    rc = launch_window(ctx, MODE_INIT, P, 1, std::max(n2, 1), n1, ctx->h_frames[frame2].n_levels);
    if (rc) return rc;
    rc = io.finish();
    if (rc) return rc;
    for (int i = 0; i < n1; i++)
        if (matches12[i] >= 0) { prev_xy[2 * i] = prev_new[2 * (size_t)i]; prev_xy[2 * i + 1] = prev_new[2 * (size_t)i + 1]; }
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

// The same with query sets that are used once (SearchBySim3's two directions, Fuse of one keyframe: LoopClosing.cc / LocalMapping
// call them with fresh projections every time): the arrays of all n_inst sets, their table and the id lists go to the device
// as ONE pinned copy with the search, nothing is uploaded beforehand or freed afterwards (see search_once above).
extern "C" int orbm_search_window_free_once(orbm_ctx *ctx, int n_inst, const int32_t *frame, const orbm_win_queries *queries,
                                            int dist_th, int chi2_gate, int32_t stride_q, int32_t *best_idx) {
    if (!ctx) return ORBM_EINVAL;
    if (n_inst < 0 || (n_inst > 0 && (!frame || !queries || !best_idx)) || stride_q < 0)
        return orbm_fail(ctx, ORBM_EINVAL, "search_window_free_once: null argument or negative size");
    if (n_inst == 0 || stride_q == 0) return ORBM_OK;
    if (n_inst > 4096) return orbm_fail(ctx, ORBM_ELIMIT, "search_window_free_once: at most 4096 query sets per call");
    int rc = orbm_check_frames(ctx, n_inst, frame, "frame");
    if (rc) return rc;
    size_t bytes = 0;
    bool ragged = false;
    for (int k = 0; k < n_inst; k++) {
        const orbm_win_queries &q = queries[k];
        if (q.n < 0 || q.n > stride_q) return orbm_fail(ctx, ORBM_EINVAL, "search_window_free_once: set %d has %d queries, stride_q is %d", k, q.n, stride_q);
        if (q.n > 0 && (!q.desc || !q.u || !q.v || !q.radius || !q.min_level || !q.max_level))
            return orbm_fail(ctx, ORBM_EINVAL, "search_window_free_once: null query array in set %d", k);
        if (chi2_gate && q.n > 0 && !q.ur) return orbm_fail(ctx, ORBM_EINVAL, "search_window_free_once: chi2_gate needs the query set's ur array");
        bytes += (size_t)q.n * 64 + 10 * 256;
        ragged = ragged || q.n != stride_q;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t out_elems = (size_t)n_inst * stride_q;
    rc = orbm_ws_reserve(ctx, bytes + (size_t)n_inst * (sizeof(QSetDev) + 8) + out_elems * 4 + 8192);
    if (rc) return rc;
    IoStage io(ctx, 0);
    std::vector<QSetDev> tab((size_t)n_inst);
    std::vector<int32_t> ids(2 * (size_t)n_inst);
    std::vector<std::vector<uint8_t>> fl((size_t)n_inst);
    static const uint8_t zero16[16] = {0};
    for (int k = 0; k < n_inst; k++) {
        const orbm_win_queries &q = queries[k];
        const size_t n = (size_t)q.n;
        QSetDev &qs = tab[k];
        memset(&qs, 0, sizeof qs);
        qs.n = q.n; qs.kind = 1;
        const void *d;
        auto put = [&](const void *p, size_t b, const void **out) { return io.in(n ? p : (const void *)zero16, std::max<size_t>(b, 16), out); };
        if ((rc = put(q.desc, n * 32, &d))) return rc; qs.desc = (const uint32_t *)d;
        if ((rc = put(q.u, n * 4, &d))) return rc; qs.f0 = (const float *)d;
        if ((rc = put(q.v, n * 4, &d))) return rc; qs.f1 = (const float *)d;
        if ((rc = put(q.radius, n * 4, &d))) return rc; qs.f2 = (const float *)d;
        if (q.ur) { if ((rc = put(q.ur, n * 4, &d))) return rc; qs.f3 = (const float *)d; }
        if ((rc = put(q.min_level, n * 4, &d))) return rc; qs.l0 = (const int32_t *)d;
        if ((rc = put(q.max_level, n * 4, &d))) return rc; qs.l1 = (const int32_t *)d;
        if (q.flags) { if ((rc = put(q.flags, n, &d))) return rc; }
        else { fl[k].assign(std::max<size_t>(n, 16), ORBM_Q_VALID | ORBM_Q_OBS); if ((rc = io.in(fl[k].data(), fl[k].size(), &d))) return rc; }
        qs.flags = (const uint8_t *)d;
        ids[k] = frame[k]; ids[(size_t)n_inst + k] = k;
    }
    const void *dtab, *dids; void *dout;
    if ((rc = io.in(tab.data(), tab.size() * sizeof(QSetDev), &dtab)) || (rc = io.in(ids.data(), ids.size() * 4, &dids)) ||
        (rc = io.out(best_idx, out_elems * 4, &dout)))
        return rc;
    if (ragged) HIPCHK(ctx, hipMemsetAsync(dout, 0xFF, out_elems * 4, ctx->stream));   // rows beyond a set's n stay -1
    WinParams P;
    memset(&P, 0, sizeof P);
    P.frames = ctx->d_frames; P.qsets = (const QSetDev *)dtab;
    P.frame_ids = (const int32_t *)dids; P.qset_ids = (const int32_t *)dids + n_inst;
    P.out = (int32_t *)dout; P.dist_th = dist_th; P.stride = stride_q;
    const int chunks = (stride_q + 255) / 256;
    hipLaunchKernelGGL(free_kernel, dim3((unsigned)(n_inst * chunks)), dim3(256), 0, ctx->stream, P, chunks, chi2_gate);
    HIPCHK(ctx, hipGetLastError());
    return io.finish();
}
