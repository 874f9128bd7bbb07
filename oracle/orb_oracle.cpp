// orb_oracle.cpp -- CPU restatement of the ORB-SLAM2 matcher hot path.  TEST INFRASTRUCTURE ONLY.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
// library; the product path (liborbmatch_hip.so and orbslam2_amd/) never does.
//
// PARITY UNPINNED: the reference tree holds no tests, golden vectors or fixtures for this path
// (SURVEY.md section 4 and 8c), and its own sources cannot be compiled here (prose lines outside
// comments, OpenCV and DBoW2 absent).  This restatement follows the reference's code lines one by
// one on flat arrays; every function cites the lines it follows.  It is pinned only by the
// hand-checkable known answers in tests/test_oracle_kat.py.
//
// Floating point: compiled with -O2 -ffp-contract=off -msse2 so that every float expression is
// evaluated in IEEE binary32 exactly in the order the reference source writes it.
//
// The flat views (orbm_frame_view, orbm_proj_queries, orbm_win_queries) are those of
// include/orbmatch.h so that a parity test hands the same arrays to both sides.

#include <math.h>
#include <stdint.h>
#include <string.h>
#include <limits.h>
#include <map>
#include <set>
#include <vector>
#include <algorithm>

#include "../include/orbmatch.h"

namespace {

const int TH_HIGH = 100;      // ORBmatcher.cc:37
const int TH_LOW = 50;        // ORBmatcher.cc:38
const int HISTO_LENGTH = 30;  // ORBmatcher.cc:39

int g_use_popcnt = 0;

// ORBmatcher.cc:2184-2200 (SWAR form, as the reference)
inline int dist_swar(const uint8_t *a, const uint8_t *b) {
    int32_t pa[8], pb[8];
    memcpy(pa, a, 32);
    memcpy(pb, b, 32);
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        unsigned int v = pa[i] ^ pb[i];
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

inline int dist_popcnt(const uint8_t *a, const uint8_t *b) {
    uint64_t pa[4], pb[4];
    memcpy(pa, a, 32);
    memcpy(pb, b, 32);
    return __builtin_popcountll(pa[0] ^ pb[0]) + __builtin_popcountll(pa[1] ^ pb[1]) +
           __builtin_popcountll(pa[2] ^ pb[2]) + __builtin_popcountll(pa[3] ^ pb[3]);
}

struct Counter {
    uint64_t n = 0;
};

inline int descriptor_distance(const uint8_t *a, const uint8_t *b, Counter &c) {
    c.n++;
    return g_use_popcnt ? dist_popcnt(a, b) : dist_swar(a, b);
}

// ORBmatcher.cc:2135-2178
void three_maxima(const int *sizes, int L, int &ind1, int &ind2, int &ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = sizes[i];
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            ind3 = ind2; ind2 = ind1; ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            ind3 = ind2; ind2 = i;
        } else if (s > max3) {
            max3 = s; ind3 = i;
        }
    }
    if (max2 < 0.1f * (float)max1) {
        ind2 = -1; ind3 = -1;
    } else if (max3 < 0.1f * (float)max1) {
        ind3 = -1;
    }
}

// ORBmatcher.cc:323-328 (and 644-649, 804-809, 1029-1034, 1911-1916, 2079-2088)
inline int rot_bin(float a1, float a2) {
    const float factor = 1.0f / HISTO_LENGTH;  // ORBmatcher.cc:235
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

struct RotHist {
    std::vector<int> bins[HISTO_LENGTH];
    void push(int bin, int v) { bins[bin].push_back(v); }
    void maxima(int &i1, int &i2, int &i3) const {
        int sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)bins[i].size();
        i1 = i2 = i3 = -1;
        three_maxima(sizes, HISTO_LENGTH, i1, i2, i3);
    }
};

// Frame.cc:506-519 + 278-298: grid[ix][iy] lists in ascending keypoint index; cell = ix*48+iy
int build_grid(const orbm_frame_view *f, uint32_t *cell_off, uint32_t *cell_idx) {
    std::vector<std::vector<uint32_t>> g(ORBM_GRID_CELLS);
    int assigned = 0;
    for (int i = 0; i < f->n; i++) {
        int posX = (int)roundf((f->x[i] - f->min_x) * f->inv_w);
        int posY = (int)roundf((f->y[i] - f->min_y) * f->inv_h);
        if (posX < 0 || posX >= ORBM_GRID_COLS || posY < 0 || posY >= ORBM_GRID_ROWS) continue;
        g[posX * ORBM_GRID_ROWS + posY].push_back((uint32_t)i);
        assigned++;
    }
    uint32_t o = 0;
    for (int c = 0; c < ORBM_GRID_CELLS; c++) {
        cell_off[c] = o;
        for (uint32_t v : g[c]) cell_idx[o++] = v;
    }
    cell_off[ORBM_GRID_CELLS] = o;
    return assigned;
}

struct Grid {
    std::vector<uint32_t> off, idx;
    explicit Grid(const orbm_frame_view *f) : off(ORBM_GRID_CELLS + 1), idx(f->n > 0 ? f->n : 1) {
        build_grid(f, off.data(), idx.data());
    }
};

// Frame.cc:434-498 (kf_mode 0) and KeyFrame.cc:766-812 (kf_mode 1: no level filter)
void features_in_area(const orbm_frame_view *f, const uint32_t *cell_off, const uint32_t *cell_idx,
                      float x, float y, float r, int minLevel, int maxLevel, int kf_mode,
                      std::vector<int> &out) {
    out.clear();
    const int nMinCellX = std::max(0, (int)floorf((x - f->min_x - r) * f->inv_w));
    if (nMinCellX >= ORBM_GRID_COLS) return;
    const int nMaxCellX = std::min((int)ORBM_GRID_COLS - 1, (int)ceilf((x - f->min_x + r) * f->inv_w));
    if (nMaxCellX < 0) return;
    const int nMinCellY = std::max(0, (int)floorf((y - f->min_y - r) * f->inv_h));
    if (nMinCellY >= ORBM_GRID_ROWS) return;
    const int nMaxCellY = std::min((int)ORBM_GRID_ROWS - 1, (int)ceilf((y - f->min_y + r) * f->inv_h));
    if (nMaxCellY < 0) return;

    const bool bCheckLevels = !kf_mode && ((minLevel > 0) || (maxLevel >= 0));

    for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const int c = ix * ORBM_GRID_ROWS + iy;
            for (uint32_t j = cell_off[c]; j < cell_off[c + 1]; j++) {
                const int k = (int)cell_idx[j];
                if (bCheckLevels) {
                    if (f->octave[k] < minLevel) continue;
                    if (maxLevel >= 0)
                        if (f->octave[k] > maxLevel) continue;
                }
                const float distx = f->x[k] - x;
                const float disty = f->y[k] - y;
                if (fabsf(distx) < r && fabsf(disty) < r) out.push_back(k);
            }
        }
    }
}

inline const uint8_t *row(const orbm_frame_view *f, int i) { return f->desc + (size_t)i * 32; }
inline uint8_t mpf(const orbm_frame_view *f, int i) { return f->mp_flags ? f->mp_flags[i] : 0; }
inline float uright(const orbm_frame_view *f, int i) { return f->uright ? f->uright[i] : -1.0f; }

// lower_bound on the sorted node-id array == std::map::lower_bound (ORBmatcher.cc:346,350)
inline int fv_lower_bound(const orbm_frame_view *f, uint32_t key) {
    return (int)(std::lower_bound(f->fv_node, f->fv_node + f->n_nodes, key) - f->fv_node);
}

// ORBmatcher.cc:179-206
bool check_dist_epipolar_line(float x1, float y1, float x2, float y2, int oct2, const float *F12,
                              const float *sigma2_2) {
    const float a = x1 * F12[0] + y1 * F12[3] + F12[6];
    const float b = x1 * F12[1] + y1 * F12[4] + F12[7];
    const float c = x1 * F12[2] + y1 * F12[5] + F12[8];
    const float num = a * x2 + b * y2 + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * sigma2_2[oct2];  // double compare, as the reference
}

}  // namespace

extern "C" {

void orc_set_popcnt(int on) { g_use_popcnt = on; }

// a1
int orc_hamming256(const uint8_t *a, const uint8_t *b) { return dist_swar(a, b); }
int orc_hamming256_popcnt(const uint8_t *a, const uint8_t *b) { return dist_popcnt(a, b); }

// a3
void orc_three_maxima(const int32_t *sizes, int L, int32_t *out3) {
    int i1 = -1, i2 = -1, i3 = -1;
    three_maxima(sizes, L, i1, i2, i3);
    out3[0] = i1; out3[1] = i2; out3[2] = i3;
}

// a4
int orc_rot_bin(float a1, float a2) { return rot_bin(a1, a2); }

// a15
int orc_build_grid(const orbm_frame_view *f, uint32_t *cell_off, uint32_t *cell_idx) {
    return build_grid(f, cell_off, cell_idx);
}

// a13 / a14; out must hold f->n ints; returns the count
int orc_features_in_area(const orbm_frame_view *f, float x, float y, float r, int minLevel,
                         int maxLevel, int kf_mode, int32_t *out) {
    Grid g(f);
    std::vector<int> v;
    features_in_area(f, g.off.data(), g.idx.data(), x, y, r, minLevel, maxLevel, kf_mode, v);
    for (size_t i = 0; i < v.size(); i++) out[i] = v[i];
    return (int)v.size();
}

// BASELINE config 2.  Scan and update rule: ORBmatcher.cc:271-300; accept: :304-308.
int orc_bf2nn(const orbm_frame_view *a, const orbm_frame_view *b, float ratio, int th, int32_t *match,
              int32_t *best_d, int32_t *second_d, uint64_t *ncmp) {
    Counter cnt;
    int nmatches = 0;
    for (int i = 0; i < a->n; i++) {
        int bestDist1 = 256, bestIdx = -1, bestDist2 = 256;
        for (int j = 0; j < b->n; j++) {
            const int dist = descriptor_distance(row(a, i), row(b, j), cnt);
            if (dist < bestDist1) {
                bestDist2 = bestDist1; bestDist1 = dist; bestIdx = j;
            } else if (dist < bestDist2) {
                bestDist2 = dist;
            }
        }
        int m = -1;
        if (bestDist1 <= th)
            if (static_cast<float>(bestDist1) < ratio * static_cast<float>(bestDist2)) { m = bestIdx; nmatches++; }
        match[i] = m;
        if (best_d) best_d[i] = bestDist1;
        if (second_d) second_d[i] = bestDist2;
    }
    if (ncmp) *ncmp = cnt.n;
    return nmatches;
}

// a7 (variant 0, ORBmatcher.cc:218-385) and a8 (variant 1, ORBmatcher.cc:703-854)
int orc_search_bow(int variant, const orbm_frame_view *A, const orbm_frame_view *B, float ratio,
                   int check_ori, int32_t *match, uint64_t *ncmp) {
    Counter cnt;
    const int nOut = variant == 0 ? B->n : A->n;
    for (int i = 0; i < nOut; i++) match[i] = -1;               // :225 / :715
    std::vector<char> claimedB(B->n > 0 ? B->n : 1, 0);          // vpMapPointMatches[idxF]!=NULL / vbMatched2
    RotHist hist;
    int nmatches = 0;
    const float *angB = (variant == 0 && B->angle_raw) ? B->angle_raw : B->angle;  // :323 uses F.mvKeys

    int ia = 0, ib = 0;
    while (ia < A->n_nodes && ib < B->n_nodes) {                  // :248 / :733
        if (A->fv_node[ia] == B->fv_node[ib]) {
            for (uint32_t pa = A->fv_off[ia]; pa < A->fv_off[ia + 1]; pa++) {
                const int idxA = (int)A->fv_idx[pa];
                const uint8_t fa = mpf(A, idxA);
                if (!(fa & ORBM_MP_PRESENT)) continue;           // :263 / :746
                if (fa & ORBM_MP_BAD) continue;                  // :266 / :748
                int bestDist1 = 256, bestIdxB = -1, bestDist2 = 256;
                for (uint32_t pb = B->fv_off[ib]; pb < B->fv_off[ib + 1]; pb++) {
                    const int idxB = (int)B->fv_idx[pb];
                    if (variant == 0) {
                        if (claimedB[idxB]) continue;            // :282
                    } else {
                        const uint8_t fb = mpf(B, idxB);
                        if (claimedB[idxB] || !(fb & ORBM_MP_PRESENT)) continue;  // :766
                        if (fb & ORBM_MP_BAD) continue;          // :769
                    }
                    const int dist = descriptor_distance(row(A, idxA), row(B, idxB), cnt);
                    if (dist < bestDist1) {
                        bestDist2 = bestDist1; bestDist1 = dist; bestIdxB = idxB;
                    } else if (dist < bestDist2) {
                        bestDist2 = dist;
                    }
                }
                const bool pass = variant == 0 ? (bestDist1 <= TH_LOW) : (bestDist1 < TH_LOW);  // :304 / :792
                if (pass) {
                    if (static_cast<float>(bestDist1) < ratio * static_cast<float>(bestDist2)) {  // :308 / :796
                        claimedB[bestIdxB] = 1;
                        if (variant == 0) match[bestIdxB] = idxA;  // :312
                        else match[idxA] = bestIdxB;               // :798-799
                        if (check_ori) {
                            const int bin = rot_bin(A->angle[idxA], angB[bestIdxB]);
                            hist.push(bin, variant == 0 ? bestIdxB : idxA);  // :330 / :811
                        }
                        nmatches++;
                    }
                }
            }
            ia++; ib++;
        } else if (A->fv_node[ia] < B->fv_node[ib]) {
            ia = fv_lower_bound(A, B->fv_node[ib]);
        } else {
            ib = fv_lower_bound(B, A->fv_node[ia]);
        }
    }
    if (check_ori) {                                            // :355-382 / :833-851
        int ind1, ind2, ind3;
        hist.maxima(ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int v : hist.bins[i]) { match[v] = -1; nmatches--; }
        }
    }
    if (ncmp) *ncmp = cnt.n;
    return nmatches;
}

// a9, ORBmatcher.cc:868-1093.  f12 row-major 3x3; epipole = (ex, ey) of :883-884.
int orc_search_triang(const orbm_frame_view *A, const orbm_frame_view *B, const float *f12,
                      const float *epipole, int only_stereo, int check_ori, int32_t *match12,
                      uint64_t *ncmp) {
    Counter cnt;
    const float ex = epipole[0], ey = epipole[1];
    for (int i = 0; i < A->n; i++) match12[i] = -1;              // :894
    RotHist hist;
    int nmatches = 0;
    int ia = 0, ib = 0;
    while (ia < A->n_nodes && ib < B->n_nodes) {
        if (A->fv_node[ia] == B->fv_node[ib]) {
            for (uint32_t pa = A->fv_off[ia]; pa < A->fv_off[ia + 1]; pa++) {
                const int idx1 = (int)A->fv_idx[pa];
                if (mpf(A, idx1) & ORBM_MP_PRESENT) continue;    // :932
                const bool bStereo1 = uright(A, idx1) >= 0;      // :937
                if (only_stereo)
                    if (!bStereo1) continue;
                int bestDist = TH_LOW;                           // :951
                int bestIdx2 = -1;
                for (uint32_t pb = B->fv_off[ib]; pb < B->fv_off[ib + 1]; pb++) {
                    const int idx2 = (int)B->fv_idx[pb];
                    if (mpf(B, idx2) & ORBM_MP_PRESENT) continue;  // :970 (vbMatched2 is never set)
                    const bool bStereo2 = uright(B, idx2) >= 0;
                    if (only_stereo)
                        if (!bStereo2) continue;
                    const int dist = descriptor_distance(row(A, idx1), row(B, idx2), cnt);
                    if (dist > TH_LOW || dist > bestDist) continue;  // :993
                    if (!bStereo1 && !bStereo2) {
                        const float distex = ex - B->x[idx2];
                        const float distey = ey - B->y[idx2];
                        if (distex * distex + distey * distey < 100 * B->scale_factors[B->octave[idx2]])
                            continue;                             // :1007
                    }
                    if (check_dist_epipolar_line(A->x[idx1], A->y[idx1], B->x[idx2], B->y[idx2],
                                                 B->octave[idx2], f12, B->level_sigma2)) {
                        bestIdx2 = idx2; bestDist = dist;         // :1016-1017
                    }
                }
                if (bestIdx2 >= 0) {
                    match12[idx1] = bestIdx2;
                    nmatches++;
                    if (check_ori) hist.push(rot_bin(A->angle[idx1], B->angle[bestIdx2]), idx1);
                }
            }
            ia++; ib++;
        } else if (A->fv_node[ia] < B->fv_node[ib]) {
            ia = fv_lower_bound(A, B->fv_node[ib]);
        } else {
            ib = fv_lower_bound(B, A->fv_node[ia]);
        }
    }
    if (check_ori) {
        int ind1, ind2, ind3;
        hist.maxima(ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int v : hist.bins[i]) { match12[v] = -1; nmatches--; }
        }
    }
    if (ncmp) *ncmp = cnt.n;
    return nmatches;
}

// a6, ORBmatcher.cc:57-166.  slot_flags (nullable => F->mp_flags) is the state of F.mvpMapPoints.
int orc_search_proj_points(const orbm_frame_view *F, const orbm_proj_queries *Q, float th, float ratio,
                           const uint8_t *slot_flags, int32_t *owner, uint64_t *ncmp) {
    Counter cnt;
    Grid g(F);
    std::vector<uint8_t> slot(F->n > 0 ? F->n : 1, 0);
    for (int i = 0; i < F->n; i++) {
        slot[i] = slot_flags ? slot_flags[i] : mpf(F, i);
        owner[i] = -1;
    }
    int nmatches = 0;
    const bool bFactor = th != 1.0;                              // :61
    std::vector<int> vIndices;
    for (int iMP = 0; iMP < Q->n; iMP++) {
        const uint8_t qf = Q->flags ? Q->flags[iMP] : (uint8_t)(ORBM_Q_VALID | ORBM_Q_OBS);
        if (!(qf & ORBM_Q_VALID)) continue;                      // :69-73
        const int nPredictedLevel = Q->level[iMP];
        float r = (Q->view_cos[iMP] > 0.998) ? 2.5f : 4.0f;      // :168-174 (double compare)
        if (bFactor) r *= th;
        features_in_area(F, g.off.data(), g.idx.data(), Q->proj_x[iMP], Q->proj_y[iMP],
                         r * F->scale_factors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel, 0,
                         vIndices);                              // :88-89
        if (vIndices.empty()) continue;
        const uint8_t *MPdescriptor = Q->desc + (size_t)iMP * 32;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int idx : vIndices) {
            if (slot[idx] & ORBM_MP_PRESENT)
                if (slot[idx] & ORBM_MP_OBS) continue;           // :113-115
            if (uright(F, idx) > 0) {
                const float er = fabsf(Q->proj_xr[iMP] - uright(F, idx));
                if (er > r * F->scale_factors[nPredictedLevel]) continue;  // :117-122
            }
            const int dist = descriptor_distance(MPdescriptor, row(F, idx), cnt);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist;
                bestLevel2 = bestLevel; bestLevel = F->octave[idx];
                bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = F->octave[idx];
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_HIGH) {                               // :153
            if (bestLevel == bestLevel2 && bestDist > ratio * bestDist2) continue;  // :157
            slot[bestIdx] = (uint8_t)(ORBM_MP_PRESENT | ((qf & ORBM_Q_OBS) ? ORBM_MP_OBS : 0));
            owner[bestIdx] = iMP;                                // :160
            nmatches++;
        }
    }
    if (ncmp) *ncmp = cnt.n;
    return nmatches;
}

// a11: the three best-1 windowed overloads in flat form (see include/orbmatch.h for the modes).
//  mode 0: ORBmatcher.cc:1845-1920 + 1926-1947;  mode 1: :2026-2092 + 2101-2120;  mode 2: :495-537.
int orc_search_window(int mode, const orbm_frame_view *F, const orbm_win_queries *Q, int dist_th,
                      int check_ori, const uint8_t *slot_flags, int32_t *owner, uint64_t *ncmp) {
    Counter cnt;
    Grid g(F);
    std::vector<uint8_t> slot(F->n > 0 ? F->n : 1, 0);
    for (int i = 0; i < F->n; i++) {
        slot[i] = slot_flags ? slot_flags[i] : mpf(F, i);
        owner[i] = -1;
    }
    RotHist hist;
    const bool use_hist = check_ori && mode != 2 && Q->angle;
    int nmatches = 0;
    std::vector<int> vIndices;
    for (int i = 0; i < Q->n; i++) {
        const uint8_t qf = Q->flags ? Q->flags[i] : (uint8_t)(ORBM_Q_VALID | ORBM_Q_OBS);
        if (!(qf & ORBM_Q_VALID)) continue;
        const float u = Q->u[i], v = Q->v[i], radius = Q->radius[i];
        const int minL = Q->min_level[i], maxL = Q->max_level[i];
        features_in_area(F, g.off.data(), g.idx.data(), u, v, radius, minL, maxL, mode == 2, vIndices);
        if (vIndices.empty()) continue;
        const uint8_t *dMP = Q->desc + (size_t)i * 32;
        int bestDist = 256, bestIdx2 = -1;
        for (int i2 : vIndices) {
            if (mode == 0) {
                if (slot[i2] & ORBM_MP_PRESENT)
                    if (slot[i2] & ORBM_MP_OBS) continue;        // :1879-1881
                if (Q->ur && uright(F, i2) > 0) {
                    const float er = fabsf(Q->ur[i] - uright(F, i2));
                    if (er > radius) continue;                   // :1883-1891
                }
            } else {
                if (slot[i2] & ORBM_MP_PRESENT) continue;        // :2048 / :512
                if (mode == 2) {
                    const int kpLevel = F->octave[i2];
                    if (kpLevel < minL || kpLevel > maxL) continue;  // :515-518
                }
            }
            const int dist = descriptor_distance(dMP, row(F, i2), cnt);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= dist_th) {                               // :1904 / :2062 / :533
            uint8_t ns = ORBM_MP_PRESENT;
            if (mode != 0 || (qf & ORBM_Q_OBS)) ns |= ORBM_MP_OBS;
            slot[bestIdx2] = ns;
            owner[bestIdx2] = i;
            nmatches++;
            if (use_hist) hist.push(rot_bin(Q->angle[i], F->angle[bestIdx2]), bestIdx2);
        }
    }
    if (use_hist) {
        int ind1, ind2, ind3;
        hist.maxima(ind1, ind2, ind3);
        for (int b = 0; b < HISTO_LENGTH; b++) {
            if (b != ind1 && b != ind2 && b != ind3)
                for (int s : hist.bins[b]) { owner[s] = -2; nmatches--; }  // -2: written, then set to NULL
        }
    }
    if (ncmp) *ncmp = cnt.n;
    return nmatches;
}

// a12: the claim-free best-1 inner loop of SearchBySim3 (:1586-1626) and Fuse (:1185-1257 with the chi-square gate,
// :1390-1421 without).  best_idx[q] = keypoint or -1 (distance above dist_th or no candidate).
int orc_search_window_free(const orbm_frame_view *F, const orbm_win_queries *Q, int dist_th, int chi2_gate,
                           int32_t *best_idx, uint64_t *ncmp) {
    Counter cnt;
    Grid g(F);
    std::vector<int> vIndices;
    for (int i = 0; i < Q->n; i++) {
        best_idx[i] = -1;
        const uint8_t qf = Q->flags ? Q->flags[i] : (uint8_t)(ORBM_Q_VALID | ORBM_Q_OBS);
        if (!(qf & ORBM_Q_VALID)) continue;
        const float u = Q->u[i], v = Q->v[i], radius = Q->radius[i];
        const int minL = Q->min_level[i], maxL = Q->max_level[i];
        features_in_area(F, g.off.data(), g.idx.data(), u, v, radius, -1, -1, 1, vIndices);
        if (vIndices.empty()) continue;
        const uint8_t *dMP = Q->desc + (size_t)i * 32;
        int bestDist = INT_MAX, bestIdx = -1;
        for (int idx : vIndices) {
            const int kpLevel = F->octave[idx];
            if (kpLevel < minL || kpLevel > maxL) continue;              // :1207 / :1406 / :1608
            if (chi2_gate) {
                const float invSigma2 = 1.0f / F->level_sigma2[kpLevel];   // mvInvLevelSigma2 (ORBextractor.cc:479)
                if (uright(F, idx) >= 0) {                                 // :1212-1225
                    const float ex = u - F->x[idx];
                    const float ey = v - F->y[idx];
                    const float er = Q->ur[i] - uright(F, idx);
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * invSigma2 > 7.8) continue;
                } else {                                                   // :1226-1238
                    const float ex = u - F->x[idx];
                    const float ey = v - F->y[idx];
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * invSigma2 > 5.99) continue;
                }
            }
            const int dist = descriptor_distance(dMP, row(F, idx), cnt);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= dist_th) best_idx[i] = bestIdx;                    // :1257 / :1421 / :1622
    }
    if (ncmp) *ncmp = cnt.n;
    return 0;
}

// a5, ORBmatcher.cc:562-689
int orc_search_init(const orbm_frame_view *F1, const orbm_frame_view *F2, float *prev_xy, int windowSize,
                    float ratio, int check_ori, int32_t *vnMatches12, uint64_t *ncmp) {
    Counter cnt;
    Grid g2(F2);
    int nmatches = 0;
    for (int i = 0; i < F1->n; i++) vnMatches12[i] = -1;          // :566
    RotHist hist;
    std::vector<int> vMatchedDistance(F2->n > 0 ? F2->n : 1, INT_MAX);  // :573
    std::vector<int> vnMatches21(F2->n > 0 ? F2->n : 1, -1);            // :574
    std::vector<int> vIndices2;
    for (int i1 = 0; i1 < F1->n; i1++) {
        const int level1 = F1->octave[i1];
        if (level1 > 0) continue;                                // :583
        features_in_area(F2, g2.off.data(), g2.idx.data(), prev_xy[2 * i1], prev_xy[2 * i1 + 1],
                         (float)windowSize, level1, level1, 0, vIndices2);  // :587
        if (vIndices2.empty()) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int i2 : vIndices2) {
            const int dist = descriptor_distance(row(F1, i1), row(F2, i2), cnt);  // :606
            if (vMatchedDistance[i2] <= dist) continue;           // :608
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2;
            } else if (dist < bestDist2) {
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_LOW) {                                // :625
            if (bestDist < (float)bestDist2 * ratio) {           // :627
                if (vnMatches21[bestIdx2] >= 0) {                // :629-633
                    vnMatches12[vnMatches21[bestIdx2]] = -1;
                    nmatches--;
                }
                vnMatches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (check_ori) hist.push(rot_bin(F1->angle[i1], F2->angle[bestIdx2]), i1);
            }
        }
    }
    if (check_ori) {                                            // :658-681
        int ind1, ind2, ind3;
        hist.maxima(ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx1 : hist.bins[i]) {
                if (vnMatches12[idx1] >= 0) { vnMatches12[idx1] = -1; nmatches--; }
            }
        }
    }
    for (int i1 = 0; i1 < F1->n; i1++)                          // :684-686
        if (vnMatches12[i1] >= 0) {
            prev_xy[2 * i1] = F2->x[vnMatches12[i1]];
            prev_xy[2 * i1 + 1] = F2->y[vnMatches12[i1]];
        }
    if (ncmp) *ncmp = cnt.n;
    return nmatches;
}

// Frame::ComputeStereoMatches, Frame.cc:632-728 (row table + descriptor stage; the SAD refinement is image-domain).
// L / R carry mvKeys / mvKeysRight in x, y, octave.  n_rows = mvImagePyramid[0].rows.  Row bands are clipped to the
// table (the reference indexes out of bounds there).
int orc_stereo_match(const orbm_frame_view *L, const orbm_frame_view *R, int n_rows, float max_d, int32_t *best_idx_r,
                     int32_t *best_dist, uint64_t *ncmp) {
    Counter cnt;
    std::vector<std::vector<size_t>> vRowIndices(n_rows, std::vector<size_t>());
    const int Nr = R->n;
    for (int iR = 0; iR < Nr; iR++) {
        const float kpY = R->y[iR];
        const float r = 2.0f * R->scale_factors[R->octave[iR]];
        const int maxr = (int)ceilf(kpY + r);
        const int minr = (int)floorf(kpY - r);
        for (int yi = minr; yi <= maxr; yi++)
            if (yi >= 0 && yi < n_rows) vRowIndices[yi].push_back(iR);
    }
    const float minD = 0;
    const float maxD = max_d;
    for (int iL = 0; iL < L->n; iL++) {
        best_idx_r[iL] = -1;
        best_dist[iL] = ORBM_TH_HIGH;
        const int levelL = L->octave[iL];
        const float vL = L->y[iL];
        const float uL = L->x[iL];
        if (!(vL >= 0) || (size_t)vL >= (size_t)n_rows) continue;
        const std::vector<size_t> &vCandidates = vRowIndices[(size_t)vL];
        if (vCandidates.empty()) continue;
        const float minU = uL - maxD;
        const float maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = ORBM_TH_HIGH;
        int bestIdxR = -1;
        for (size_t iC = 0; iC < vCandidates.size(); iC++) {
            const size_t iR = vCandidates[iC];
            if (R->octave[iR] < levelL - 1 || R->octave[iR] > levelL + 1) continue;
            const float uR = R->x[iR];
            if (uR >= minU && uR <= maxU) {
                const int dist = orc_hamming256(L->desc + 32 * (size_t)iL, R->desc + 32 * iR);
                cnt.n++;
                if (dist < bestDist) {
                    bestDist = dist;
                    bestIdxR = (int)iR;
                }
            }
        }
        best_idx_r[iL] = bestIdxR;
        best_dist[iL] = bestDist;
    }
    if (ncmp) *ncmp = cnt.n;
    return 0;
}

// MapPoint::ComputeDistinctiveDescriptors, MapPoint.cc:408-441: index of the chosen descriptor among n rows
int orc_distinctive_descriptor(const uint8_t *desc, int n) {
    if (n <= 0) return -1;                                              // :405-406
    const size_t N = (size_t)n;
    std::vector<float> Distances(N * N);                                // float Distances[N][N]  (:412)
    for (size_t i = 0; i < N; i++) {
        Distances[i * N + i] = 0;
        for (size_t j = i + 1; j < N; j++) {
            int distij = orc_hamming256(desc + 32 * i, desc + 32 * j);
            Distances[i * N + j] = distij;
            Distances[j * N + i] = distij;
        }
    }
    int BestMedian = INT_MAX;
    int BestIdx = 0;
    for (size_t i = 0; i < N; i++) {
        std::vector<int> vDists(Distances.begin() + i * N, Distances.begin() + (i + 1) * N);
        std::sort(vDists.begin(), vDists.end());
        int median = vDists[0.5 * (N - 1)];                             // :434
        if (median < BestMedian) {
            BestMedian = median;
            BestIdx = (int)i;
        }
    }
    return BestIdx;
}

// DBoW2 (third-party dependency of the reference, absent from its tree; parity unpinned): TemplatedVocabulary<FORB::TDescriptor,
// FORB>::transform(features, BowVector&, FeatureVector&, levelsup), word / node part, as Frame.cc:526-533 and KeyFrame.cc:69-78
// call it.  The tree is rebuilt from the flat view into DBoW2's own shape (a vector of nodes with child-id vectors) and the
// FeatureVector is the std::map<NodeId, std::vector<unsigned>> that FeatureVector::addFeature fills.
int orc_transform(const orbm_vocab_view *V, const uint8_t *desc, int n, int levelsup, int32_t *word_out, int32_t *node_out,
                  uint32_t *fv_node, uint32_t *fv_off, uint32_t *fv_idx, int32_t *bv_n, uint32_t *bv_word, double *bv_value) {
    struct Node {
        std::vector<uint32_t> children;
        const uint8_t *descriptor;
        double weight;
        int32_t word_id;
        bool isLeaf() const { return children.empty(); }
    };
    std::vector<Node> m_nodes(V->n_nodes);
    for (int i = 0; i < V->n_nodes; i++) {
        Node &nd = m_nodes[i];
        nd.children.assign(V->child_idx + V->child_off[i], V->child_idx + V->child_off[i + 1]);
        nd.descriptor = V->desc + 32 * (size_t)i;
        nd.weight = V->weight ? V->weight[i] : 1.0;
        nd.word_id = V->word_id[i];
    }
    const int m_L = V->depth;
    // BowVector part of transform(): LNorm norm; bool must = m_scoring_object->mustNormalize(norm);
    const bool tf = V->weighting == 0 || V->weighting == 1;                  // TF_IDF, TF
    const bool must = V->scoring != 5;                                       // every scoring but DOT_PRODUCT normalises
    const bool l2 = V->scoring == 1;                                         // L2_NORM -> L2, the others -> L1
    std::map<uint32_t, double> v;
    std::map<uint32_t, std::vector<unsigned int>> fv;
    for (int i_feature = 0; i_feature < n; i_feature++) {
        const uint8_t *feature = desc + 32 * (size_t)i_feature;
        // transform(feature, word_id, weight, &nid, levelsup)
        std::vector<uint32_t> nodes;
        const int nid_level = m_L - levelsup;
        uint32_t nid = 0;
        if (nid_level <= 0) nid = 0;   // root
        uint32_t final_id = 0;         // root
        int current_level = 0;
        do {
            ++current_level;
            nodes = m_nodes[final_id].children;
            final_id = nodes[0];
            double best_d = orc_hamming256(feature, m_nodes[final_id].descriptor);
            for (auto nit = nodes.begin() + 1; nit != nodes.end(); ++nit) {
                uint32_t id = *nit;
                double d = orc_hamming256(feature, m_nodes[id].descriptor);
                if (d < best_d) {
                    best_d = d;
                    final_id = id;
                }
            }
            if (current_level == nid_level) nid = final_id;
        } while (!m_nodes[final_id].isLeaf());
        int32_t word_id = m_nodes[final_id].word_id;
        double w = m_nodes[final_id].weight;
        if (word_out) word_out[i_feature] = word_id;
        if (node_out) node_out[i_feature] = w > 0 ? (int32_t)nid : -1;
        if (w > 0) {   // not stopped
            {   // v.addWeight(id, w)  /  v.addIfNotExist(id, w)
                auto bit = v.lower_bound((uint32_t)word_id);
                if (bit != v.end() && bit->first == (uint32_t)word_id) {
                    if (tf) bit->second += w;
                } else {
                    v.insert(bit, std::make_pair((uint32_t)word_id, w));
                }
            }
            // FeatureVector::addFeature(nid, i_feature)
            auto vit = fv.lower_bound(nid);
            if (vit != fv.end() && vit->first == nid)
                vit->second.push_back(i_feature);
            else {
                vit = fv.insert(vit, std::make_pair(nid, std::vector<unsigned int>()));
                vit->second.push_back(i_feature);
            }
        }
    }
    if (tf && !v.empty() && !must) {   // unnecessary when normalizing
        const double nd = v.size();
        for (auto &kv : v) kv.second /= nd;
    }
    if (must) {                        // BowVector::normalize(norm)
        double norm = 0.0;
        if (!l2) {
            for (auto &kv : v) norm += fabs(kv.second);
        } else {
            for (auto &kv : v) norm += kv.second * kv.second;
            norm = sqrt(norm);
        }
        if (norm > 0.0)
            for (auto &kv : v) kv.second /= norm;
    }
    if (bv_n) {
        int t = 0;
        for (auto &kv : v) { bv_word[t] = kv.first; bv_value[t] = kv.second; t++; }
        *bv_n = t;
    }
    int j = 0;
    uint32_t pos = 0;
    for (auto &kv : fv) {
        fv_node[j] = kv.first;
        fv_off[j] = pos;
        for (unsigned int i : kv.second) fv_idx[pos++] = i;
        j++;
    }
    fv_off[j] = pos;
    return j;
}

// DBoW2 L1Scoring::score(v1, v2) on std::map BowVectors (third-party, absent from the reference tree; parity unpinned):
// the walk with lower_bound jumps, the double sum in ascending word order, -score / 2.  Also counts the common words.
static double l1_score_maps(const std::map<uint32_t, double> &v1, const std::map<uint32_t, double> &v2, int *common) {
    auto v1_it = v1.begin(), v2_it = v2.begin();
    const auto v1_end = v1.end(), v2_end = v2.end();
    double score = 0;
    int n = 0;
    while (v1_it != v1_end && v2_it != v2_end) {
        const double &vi = v1_it->second;
        const double &wi = v2_it->second;
        if (v1_it->first == v2_it->first) {
            score += fabs(vi - wi) - fabs(vi) - fabs(wi);
            n++;
            ++v1_it;
            ++v2_it;
        } else if (v1_it->first < v2_it->first) {
            v1_it = v1.lower_bound(v2_it->first);
        } else {
            v2_it = v2.lower_bound(v1_it->first);
        }
    }
    if (common) *common = n;
    score = -score / 2.0;
    return score;
}

static std::map<uint32_t, double> to_map(const orbm_bowvec_view *v) {
    std::map<uint32_t, double> m;
    for (int i = 0; i < v->n; i++) m[v->word[i]] = v->value[i];
    return m;
}

double orc_bow_score(const orbm_bowvec_view *a, const orbm_bowvec_view *b, int32_t *common) {
    int n = 0;
    const double s = l1_score_maps(to_map(a), to_map(b), &n);
    if (common) *common = n;
    return s;
}

// KeyFrameDatabase::DetectLoopCandidates (KeyFrameDataBase.cc:109-285; is_reloc = 0) and DetectRelocalizationCandidates
// (:287-431; is_reloc = 1) on keyframe INDICES.  db holds, per keyframe, its BowVector, its mnId, its ten best covisible
// keyframes in order (GetBestCovisibilityKeyFrames(10)) and the persistent bookkeeping fields the reference keeps in
// KeyFrame (query id, word counter, last score); inv_* is the inverted file in list order.  connected[k] != 0: keyframe
// k is in the query's GetConnectedKeyFrames() (loop only).  Returns the number of candidates written to out.
struct orc_kfdb {
    int32_t n_kf, n_words;
    const orbm_bowvec_view *bow;
    const int64_t *mn_id;
    const int32_t *nb_off, *nb_idx;
    const int32_t *inv_off, *inv_idx;
    int64_t *query;      // mnLoopQuery / mnRelocQuery
    int32_t *words;      // mnLoopWords / mnRelocWords
    float *score;        // mLoopScore / mRelocScore
};

int orc_detect_candidates(int is_reloc, const orc_kfdb *db, const orbm_bowvec_view *q_bow, int64_t q_id, const uint8_t *connected,
                          float minScore, int32_t *out) {
    std::vector<int> lKFsSharingWords;                                       // :125-146 / :305-325
    for (int w = 0; w < q_bow->n; w++) {
        const uint32_t word = q_bow->word[w];
        for (int32_t j = db->inv_off[word]; j < db->inv_off[word + 1]; j++) {
            const int k = db->inv_idx[j];
            if (db->query[k] != q_id) {
                db->words[k] = 0;
                if (is_reloc || !connected[k]) {
                    db->query[k] = q_id;
                    lKFsSharingWords.push_back(k);
                }
            }
            db->words[k]++;
        }
    }
    if (lKFsSharingWords.empty()) return 0;
    int maxCommonWords = 0;                                                  // :160-170 / :332-338
    for (int k : lKFsSharingWords)
        if (db->words[k] > maxCommonWords) maxCommonWords = db->words[k];
    int minCommonWords = maxCommonWords * 0.8f;
    const std::map<uint32_t, double> qmap = to_map(q_bow);
    std::vector<std::pair<float, int>> lScoreAndMatch;                       // :179-196 / :347-361
    for (int k : lKFsSharingWords)
        if (db->words[k] > minCommonWords) {
            float si = (float)l1_score_maps(qmap, to_map(&db->bow[k]), nullptr);
            db->score[k] = si;
            if (is_reloc || si >= minScore) lScoreAndMatch.push_back(std::make_pair(si, k));
        }
    if (lScoreAndMatch.empty()) return 0;
    std::vector<std::pair<float, int>> lAccScoreAndMatch;                    // :205-232 / :370-400
    float bestAccScore = is_reloc ? 0.f : minScore;
    for (auto &sm : lScoreAndMatch) {
        const int k = sm.second;
        float bestScore = sm.first;
        float accScore = sm.first;
        int best = k;
        for (int32_t j = db->nb_off[k]; j < db->nb_off[k + 1]; j++) {
            const int k2 = db->nb_idx[j];
            if (is_reloc) {
                if (db->query[k2] != q_id) continue;
            } else if (!(db->query[k2] == q_id && db->words[k2] > minCommonWords)) {
                continue;
            }
            accScore += db->score[k2];
            if (db->score[k2] > bestScore) {
                best = k2;
                bestScore = db->score[k2];
            }
        }
        lAccScoreAndMatch.push_back(std::make_pair(accScore, best));
        if (accScore > bestAccScore) bestAccScore = accScore;
    }
    float minScoreToRetain = 0.75f * bestAccScore;                           // :241-262 / :403-431
    std::set<int> already;
    int n_out = 0;
    for (auto &am : lAccScoreAndMatch)
        if (am.first > minScoreToRetain && !already.count(am.second)) {
            out[n_out++] = am.second;
            already.insert(am.second);
        }
    return n_out;
}

// MapPoint.cc:569-602 (PredictScale) in flat form: host-side helper of the a11 gates.  MapPoint.cc sees
// `using namespace std` through the DBoW2 headers, so log(float)/ceil(float) resolve to the float overloads.
int orc_predict_scale(float max_distance, float current_dist, float log_scale_factor, int n_levels) {
    float ratio = max_distance / current_dist;
    int nScale = (int)ceilf(logf(ratio) / log_scale_factor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= n_levels) nScale = n_levels - 1;
    return nScale;
}

}  // extern "C"
