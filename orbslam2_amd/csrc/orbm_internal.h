// orbm_internal.h -- shared declarations of liborbmatch_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/orbmatch.h"

// Device-resident layout of one Frame / KeyFrame (replaces Frame.h:235-291 / KeyFrame.h:232-282):
//  desc      packed 32-byte rows, 256-byte aligned base, row i = keypoint i
//  x,y,...   keypoint SoA
//  cell_*    64x48 grid as CSR, cell = ix*48+iy, indices ascending inside a cell (Frame.cc:278-298);
//            cell_rec[j] repeats keypoint cell_idx[j]'s x, y, uright, index and octave in CSR order
//  fv_*      DBoW2::FeatureVector as CSR (nodes ascending)
struct FrameDev {
    int32_t n, n_nodes, n_levels, pad0;
    const uint32_t *desc;
    const float *x, *y, *angle, *angle_raw, *uright;
    const uint8_t *octave;
    uint8_t *mp_flags;
    uint32_t *cell_off, *cell_idx;
    float4 *cell_rec;   // cell-ordered copy {x, y, uright, idx | octave << 16 | iy << 20 | ix << 26}: the window walk streams it
    // cell_nxt[j * 8 + p], j in [0, T] (T = cell_off[3072] keypoints are in the grid): first position j' >= j of the
    // cell-ordered stream whose octave is p - 1 or p (T if none): a walk that only accepts octaves {p-1, p} hops over the other records.  Null when n_levels > 8 or
    // n > 8192.
    uint16_t *cell_nxt;
    const uint32_t *fv_node, *fv_off, *fv_idx;
    // every descriptor bit as one int8 (+1 for 0, -1 for 1), 256 B per row, rows padded with zeros to a multiple of 32:
    // the operand image of the MFMA form of the brute-force search.  Built on first use (orbm_ensure_desc8), else null.
    const uint32_t *desc8;
    // the same as FP4 (E2M1) values, two per byte: +1.0 = 0x2, -1.0 = 0xA; 128 B per row (operands of the MX-scaled MFMA form)
    const uint32_t *desc4;
    // the descriptors again (32-byte rows) in FeatureVector order: row j = keypoint fv_idx[j], so a vocabulary node's rows are
    // consecutive and 32 more zero rows follow the last one.  Operand image of the matrix-core form of the BoW searches
    // (bowm.hip); built on first use, dropped when the FeatureVector is rebuilt.
    const uint32_t *descfv;
    float min_x, min_y, max_x, max_y, inv_w, inv_h;
    float sf[ORBM_MAX_LEVELS];
    float sigma2[ORBM_MAX_LEVELS];
};

// Device-resident query set: orbm_proj_queries (kind 0) or orbm_win_queries (kind 1).
struct QSetDev {
    int32_t n, kind;
    const uint32_t *desc;
    const float *f0, *f1, *f2, *f3;   // kind0: proj_x, proj_y, proj_xr, view_cos | kind1: u, v, radius, ur(null ok)
    const float *angle;               // kind1 only, may be null
    const int32_t *l0, *l1;           // kind0: level, - | kind1: min_level, max_level
    const uint8_t *flags;
    // kind0 only, and then the ONLY per-query parameters on the device (f0..f3, l0, flags are null): 16 bytes per query
    // {proj_x, proj_y, proj_xr, level | flags << 8 | (view_cos >= 0.998f) << 16}, so that a query gathered by index costs
    // one load instead of six
    const uint32_t *pk;
};

struct Arena {
    void *base = nullptr;
    size_t bytes = 0;   // bytes in use
    size_t cap = 0;     // bytes allocated
    int live = 0;       // objects still referencing it
};

// Device-resident vocabulary tree (DBoW2 TemplatedVocabulary): node 0 is the root.  node_rec[i] = {x, y}: y & 0x7FFFFFFF =
// number of children (0 = leaf); y bit 31 set: the children have consecutive ids starting at x (breadth-first files,
// the common case: no index load); clear: they are child_idx[x .. x + count) in stored order.  32-byte descriptor per
// node, word id per leaf (-1 inside; bit 31 = the word's weight is not positive).
struct VocabDev {
    int32_t n_nodes = 0, depth = 0;
    const uint2 *node_rec = nullptr;
    const uint32_t *child_idx = nullptr, *desc = nullptr;
    const int32_t *word_id = nullptr;
    const double *weight = nullptr;     // per node
    int32_t weighting = 0, scoring = 0; // DBoW2 WeightingType / ScoringType
    Arena arena;
};

// Device-resident DBoW2::BowVector: ascending word ids and their values
struct BowVecDev {
    int32_t n;
    const uint32_t *word;
    const double *value;
};

struct orbm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    std::string err;
    hipDeviceProp_t prop;
    // measurement (orbm_kernel_timing): event pairs around the dominant kernel of the brute-force search
    std::vector<hipEvent_t> tev;     // 2 per launch
    int tev_used = 0;
    // frame table
    std::vector<FrameDev> h_frames;
    std::vector<int> frame_arena;      // arena index per slot, -1 = free slot
    std::vector<int> frame_arena8;     // arena of the slot's desc8 image, -1 = none
    std::vector<int> frame_arena4;     // arena of the slot's desc4 image, -1 = none
    std::vector<int> frame_arena4f;    // arena of the slot's descfv image, -1 = none
    std::vector<int32_t> free_frames;
    FrameDev *d_frames = nullptr;
    size_t d_frames_cap = 0;
    // query-set table
    std::vector<QSetDev> h_qsets;
    std::vector<int> qset_arena;
    std::vector<int32_t> free_qsets;
    QSetDev *d_qsets = nullptr;
    size_t d_qsets_cap = 0;
    std::vector<Arena> arenas;
    std::vector<VocabDev> vocabs;      // slot free when arena.base == nullptr
    // BowVector table (bowscore.hip): one arena per upload batch
    std::vector<BowVecDev> h_bvecs;
    std::vector<int> bvec_arena;       // index into bvec_arenas, -1 = free slot
    std::vector<int32_t> free_bvecs;
    std::vector<Arena> bvec_arenas;
    BowVecDev *d_bvecs = nullptr;
    size_t d_bvecs_cap = 0;
    // released arenas kept for reuse: per-call uploads of the drop-in class (a frame, a query set) would otherwise
    // pay a hipMalloc + hipFree each; bounded in count and bytes
    std::vector<Arena> arena_pool;
    size_t arena_pool_bytes = 0;
    // grow-only workspace for staged host I/O and kernel scratch
    void *ws = nullptr;
    size_t ws_bytes = 0;
    // pinned host mirrors: `pin` shadows ws (host-mode index lists and outputs travel as ONE copy each way),
    // `pin_up` stages uploads.  Pageable copies cost ~70 us each on this platform, pinned ones ~10.
    void *pin = nullptr;
    size_t pin_bytes = 0;
    void *pin_up = nullptr;
    size_t pin_up_bytes = 0;
    void *ws2 = nullptr;   // kernel scratch (separate from I/O staging)
    size_t ws2_bytes = 0;
    void *ws3 = nullptr;   // orbm_allpairs_batch: match rows + counts of the batch (they never leave the device)
    size_t ws3_bytes = 0;
};

int orbm_fail(orbm_ctx *ctx, int code, const char *fmt, ...);
#define HIPCHK(ctx, expr)                                                                     \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess)                                                                \
            return orbm_fail(ctx, ORBM_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                             __FILE__, __LINE__);                                             \
    } while (0)

int orbm_arena_alloc(orbm_ctx *ctx, size_t bytes, Arena *out);   // from the pool or hipMalloc
void orbm_arena_release(orbm_ctx *ctx, Arena &a);                // to the pool or hipFree
int orbm_ws_reserve(orbm_ctx *ctx, size_t bytes);    // I/O staging
void *orbm_pin_up(orbm_ctx *ctx, size_t bytes);      // pinned upload staging of at least `bytes`, or null (too large / failed)
int orbm_ws2_reserve(orbm_ctx *ctx, size_t bytes);   // kernel scratch
hipEvent_t *orbm_timing_slot(orbm_ctx *ctx);         // next event pair of orbm_kernel_timing, or null
int orbm_check_frames(orbm_ctx *ctx, int n, const int32_t *ids, const char *what);
int orbm_ensure_image(orbm_ctx *ctx, int bits, int n, const int32_t *ids);   // bits = 8 (desc8), 4 (desc4) or 5 (descfv, raw rows in FeatureVector order); ids == nullptr: every live frame
void orbm_drop_fv_image(orbm_ctx *ctx, int32_t id);                          // host side of "the FeatureVector of frame id was rebuilt" (the device entry is cleared by the rebuilding kernel)
int orbm_bvecs_reserve(orbm_ctx *ctx, int n_vecs, const int32_t *cap, const int32_t *n_init, int32_t *out_ids, void **arena_base,
                       std::vector<size_t> &o_word, std::vector<size_t> &o_val, size_t *total_bytes);   // bowscore.hip
int orbm_check_qsets(orbm_ctx *ctx, int n, const int32_t *ids, int kind, const char *what);
extern "C" int orbm_bf2nn_prepare(orbm_ctx *ctx);
int orbm_bow_cap(float ratio);                        // bowm.hip: distances below it are the only ones that can change a SearchByBoW result
int orbm_launch_bowm(orbm_ctx *ctx, int mode, int n_pairs, const int32_t *dfa, const int32_t *dfb, float ratio, int capd,
                     int check_ori, int stride, int32_t *dm, int32_t *dc, int splits, int *pair_acc, int only_stereo,
                     const float *df12, const float *dep);   // bowm.hip: MFMA form; splits > 1: that many workgroups per pair, accumulators in pair_acc
int orbm_bow_scratch_resolve(orbm_ctx *ctx);         // bow.hip: device address of the static scratch array of ctx->device (orbm_create calls it)

// hipFuncAttributeMaxDynamicSharedMemorySize stays with the function once set: ask the runtime only when a launch needs more
// dynamic LDS than any launch of that kernel before it on this device (two runtime calls less per windowed search).
#ifdef __HIPCC__
#include <atomic>
template <auto Kernel>
inline hipError_t orbm_allow_dynamic_lds(const orbm_ctx *ctx, size_t bytes) {
    static std::atomic<size_t> allowed[64];
    std::atomic<size_t> &a = allowed[(unsigned)ctx->device & 63u];
    if (bytes <= 48 * 1024 || bytes <= a.load(std::memory_order_relaxed)) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) a.store(bytes, std::memory_order_relaxed);
    return e;
}
#endif

// Helper that stages host index lists / outputs through the I/O workspace when io_on_device == 0.
struct IoStage {
    orbm_ctx *ctx;
    bool dev;
    size_t used = 0;
    size_t in_end = 0;        // [0, in_end) of the workspace holds inputs not yet copied to the device
    bool in_sent = false;
    size_t out_begin = 0;
    struct Out { void *host; void *devp; size_t bytes; };
    std::vector<Out> outs;
    IoStage(orbm_ctx *c, int io_on_device) : ctx(c), dev(io_on_device != 0) {}
    // total bytes must have been reserved with orbm_ws_reserve before the first in()/out()
    void *carve(size_t bytes);
    int in(const void *p, size_t bytes, const void **devp);          // copies host->device if needed
    int out(void *p, size_t bytes, void **devp);                     // registers device->host copy
    int send_inputs();                                               // one H2D copy of everything in() staged
    int finish();                                                    // copies outputs back + sync (host mode)
};

// ---- device helpers (gfx950) -------------------------------------------------------------------
#ifdef __HIPCC__
// A pointer that device code reads from a table in memory is a "flat" pointer to the compiler: every access through it is
// a flat_load (counts on both wait counters, so it also holds up LDS reads, and takes no scalar base).  Typed as a
// pointer into the global address space the accesses become global_loads.  FrameG / QSetG are FrameDev / QSetDev with
// such pointers: same member names, built once at the top of a kernel (members a kernel does not use cost nothing).
#define ORBM_GLOBAL __attribute__((address_space(1)))
typedef uint32_t v4u __attribute__((ext_vector_type(4)));   // 16-byte loads through such pointers (HIP's uint4 / float4 are
typedef float v4f __attribute__((ext_vector_type(4)));      // classes whose copy constructors take flat references)
template <class T>
using gptr = ORBM_GLOBAL T *;
template <class T>
__device__ __forceinline__ gptr<T> in_global(T *p) {
    return (gptr<T>)p;
}
struct FrameG {
    int32_t n, n_nodes, n_levels;
    gptr<const uint32_t> desc;
    gptr<const float> x, y, angle, angle_raw, uright;
    gptr<const uint8_t> octave;
    gptr<uint8_t> mp_flags;
    gptr<uint32_t> cell_off, cell_idx;
    gptr<const v4f> cell_rec;
    gptr<uint16_t> cell_nxt;
    gptr<const uint32_t> fv_node, fv_off, fv_idx, desc8, desc4, descfv;
    float min_x, min_y, max_x, max_y, inv_w, inv_h;
    gptr<const float> sf, sigma2;
    FrameG() = default;
    __device__ __forceinline__ explicit FrameG(const FrameDev &f)
        : n(f.n), n_nodes(f.n_nodes), n_levels(f.n_levels), desc(in_global(f.desc)), x(in_global(f.x)), y(in_global(f.y)),
          angle(in_global(f.angle)), angle_raw(in_global(f.angle_raw)), uright(in_global(f.uright)), octave(in_global(f.octave)),
          mp_flags(in_global(f.mp_flags)), cell_off(in_global(f.cell_off)), cell_idx(in_global(f.cell_idx)),
          cell_rec((gptr<const v4f>)f.cell_rec), cell_nxt(in_global(f.cell_nxt)), fv_node(in_global(f.fv_node)),
          fv_off(in_global(f.fv_off)), fv_idx(in_global(f.fv_idx)), desc8(in_global(f.desc8)), desc4(in_global(f.desc4)), descfv(in_global(f.descfv)),
          min_x(f.min_x), min_y(f.min_y), max_x(f.max_x), max_y(f.max_y), inv_w(f.inv_w), inv_h(f.inv_h),
          sf(in_global(&f.sf[0])), sigma2(in_global(&f.sigma2[0])) {}
};
struct QSetG {
    int32_t n, kind;
    gptr<const uint32_t> desc;
    gptr<const float> f0, f1, f2, f3, angle;
    gptr<const int32_t> l0, l1;
    gptr<const uint8_t> flags;
    gptr<const uint32_t> pk;
    QSetG() = default;
    __device__ __forceinline__ explicit QSetG(const QSetDev &q)
        : n(q.n), kind(q.kind), desc(in_global(q.desc)), f0(in_global(q.f0)), f1(in_global(q.f1)), f2(in_global(q.f2)),
          f3(in_global(q.f3)), angle(in_global(q.angle)), l0(in_global(q.l0)), l1(in_global(q.l1)), flags(in_global(q.flags)), pk(in_global(q.pk)) {}
};

__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// 256-bit Hamming distance: 8 x (v_xor_b32 + v_bcnt_u32_b32 with accumulate)  [ORBmatcher.cc:2184-2200]
__device__ __forceinline__ uint32_t hamming256(const uint32_t (&a)[8], const uint32_t (&b)[8]) {
    // the accumulate chain is written as asm: left to itself hipcc re-associates the eight adds into
    // 8 x v_bcnt(x, 0) + 3 x v_add3_u32, i.e. 19 instead of 16 VALU ops per comparison.
    uint32_t d = __builtin_popcount(a[0] ^ b[0]);
#pragma unroll
    for (int i = 1; i < 8; i++) {
        const uint32_t x = a[i] ^ b[i];
        asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d) : "v"(x));
    }
    return d;
}

// Blocks b and b+8 share an XCD (round-robin dispatch): remap so that consecutive logical ids
// share an L2.  Bijective for any grid size (cdna_hip_programming.md T1).
__device__ __forceinline__ uint32_t xcd_remap(uint32_t orig, uint32_t nwg) {
    const uint32_t q = nwg >> 3, r = nwg & 7, xcd = orig & 7, k = orig >> 3;
    const uint32_t base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

// rotation-histogram bin, ORBmatcher.cc:323-328
__device__ __forceinline__ int rot_bin(float a1, float a2) {
    const float factor = 1.0f / ORBM_HISTO_LENGTH;
    float rot = __fsub_rn(a1, a2);
    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
    int bin = (int)roundf(__fmul_rn(rot, factor));
    if (bin == ORBM_HISTO_LENGTH) bin = 0;
    return bin;
}

// ORBmatcher::CheckDistEpipolarLine, ORBmatcher.cc:179-206 (float products and sums in source order, the compare in double)
__device__ __forceinline__ bool check_dist_epipolar_line(float x1, float y1, float x2, float y2, float sigma2,
                                                         const float *F12) {
    const float a = __fadd_rn(__fadd_rn(__fmul_rn(x1, F12[0]), __fmul_rn(y1, F12[3])), F12[6]);
    const float b = __fadd_rn(__fadd_rn(__fmul_rn(x1, F12[1]), __fmul_rn(y1, F12[4])), F12[7]);
    const float c = __fadd_rn(__fadd_rn(__fmul_rn(x1, F12[2]), __fmul_rn(y1, F12[5])), F12[8]);
    const float num = __fadd_rn(__fadd_rn(__fmul_rn(a, x2), __fmul_rn(b, y2)), c);
    const float den = __fadd_rn(__fmul_rn(a, a), __fmul_rn(b, b));
    if (den == 0) return false;
    const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
    return (double)dsqr < __dmul_rn(3.84, (double)sigma2);
}

// ComputeThreeMaxima, ORBmatcher.cc:2135-2178 (single lane)
__device__ __forceinline__ void three_maxima(const int *sizes, int &ind1, int &ind2, int &ind3) {
    // (size << 8 | 255 - bin) keys: a larger size ranks first, the lower bin index among equal sizes (strict '>' in a
    // scan by ascending bin); sizes of 0 never rank (the reference starts from max = 0 with strict '>')
    // fully unrolled: the 30 reads are issued together (a rolled loop waits for each one: 2 us instead of 0.3)
    uint32_t k1 = 0, k2 = 0, k3 = 0;
#pragma unroll
    for (int i = 0; i < ORBM_HISTO_LENGTH; i++) {
        const int s = sizes[i];
        uint32_t x = s > 0 ? ((uint32_t)s << 8) | (uint32_t)(255 - i) : 0u, hi;
        hi = max(k1, x); x = min(k1, x); k1 = hi;
        hi = max(k2, x); x = min(k2, x); k2 = hi;
        k3 = max(k3, x);
    }
    const int max1 = (int)(k1 >> 8), max2 = (int)(k2 >> 8), max3 = (int)(k3 >> 8);
    int a1 = k1 ? 255 - (int)(k1 & 255u) : -1, a2 = k2 ? 255 - (int)(k2 & 255u) : -1, a3 = k3 ? 255 - (int)(k3 & 255u) : -1;
    if ((float)max2 < __fmul_rn(0.1f, (float)max1)) {
        a2 = -1; a3 = -1;
    } else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) {
        a3 = -1;
    }
    ind1 = a1; ind2 = a2; ind3 = a3;
}
#endif
