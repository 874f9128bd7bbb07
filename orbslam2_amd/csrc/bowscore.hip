// bowscore.hip -- SURVEY 8(f) rank 2, the arithmetic of KeyFrameDatabase::DetectLoopCandidates /
// DetectRelocalizationCandidates (KeyFrameDataBase.cc:109-285, 287-431): `mpVoc->score(query.mBowVec, pKFi->mBowVec)` for
// every keyframe that shares enough words (:188, :356), batched, plus the number of shared words (mnLoopWords /
// mnRelocWords, :140, :322).  DBoW2's L1Scoring::score (ORBvoc.txt's scoring; DBoW2 is absent from the reference tree):
//     score = sum over common words, ascending word id, of |vi - wi| - |vi| - |wi|   (double, in that order);  return -score / 2
// One wave per (query, keyframe) pair: the query's word ids sit in LDS, each lane takes one of the keyframe's words,
// finds it by binary search, and the wave adds the hits' terms one by one in word order (the double sum is sequential
// in the reference, so it is here: identical bits).  The list logic around it (inverted file order, covisibility
// accumulation) is pointer work and stays on the host (host/KeyFrameDatabase.cc).
#include <algorithm>
#include <string.h>
#include "orbm_internal.h"

namespace {

constexpr int BVEC_MAX_WORDS = 16384;      // query words staged in LDS (64 KB)

__global__ __launch_bounds__(64) void bow_score_kernel(const BowVecDev *__restrict__ table, const int32_t *__restrict__ va,
                                                       const int32_t *__restrict__ vb, double *__restrict__ score,
                                                       int32_t *__restrict__ common) {
    extern __shared__ uint32_t qw[];
    const int p = blockIdx.x, lane = threadIdx.x;
    struct { int32_t n; gptr<const uint32_t> word; gptr<const double> value; } A, B;    // pointers from a table: typed global
    A.n = table[va[p]].n; A.word = in_global(table[va[p]].word); A.value = in_global(table[va[p]].value);
    B.n = table[vb[p]].n; B.word = in_global(table[vb[p]].word); B.value = in_global(table[vb[p]].value);
    for (int i = lane; i < A.n; i += 64) qw[i] = A.word[i];
    __syncthreads();
    double acc = 0.0;
    int ncommon = 0;
    for (int base = 0; base < B.n; base += 64) {
        const int i = base + lane;
        const bool have = i < B.n;
        const uint32_t w = have ? B.word[i] : 0xFFFFFFFFu;
        int lo = 0, hi = A.n;
        while (lo < hi) {                                  // lower_bound(w) in the query's words
            const int mid = (lo + hi) >> 1;
            if (qw[mid] < w) lo = mid + 1; else hi = mid;
        }
        const bool hit = have && lo < A.n && qw[lo] == w;
        double t = 0.0;
        if (hit) {
            const double vi = A.value[lo], wi = B.value[i];
            t = fabs(vi - wi) - fabs(vi) - fabs(wi);
        }
        unsigned long long mask = __ballot(hit);
        ncommon += __popcll(mask);
        while (mask) {                                     // wave-uniform: the hits' terms in ascending word order
            const int l = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            // l is wave-uniform: two v_readlane_b32 instead of a cross-lane permute through LDS
            acc += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(t), l), __builtin_amdgcn_readlane(__double2loint(t), l));
        }
    }
    if (lane == 0) {
        score[p] = -acc / 2.0;
        if (common) common[p] = ncommon;
    }
}

}  // namespace

// Table entries + one arena for n_vecs BowVectors of the given capacities; entry k starts with n_init[k] words (the
// caller fills the arrays: copy for an upload, a kernel for ComputeBoW).  o_word / o_val: byte offsets into the arena.
int orbm_bvecs_reserve(orbm_ctx *ctx, int n_vecs, const int32_t *cap, const int32_t *n_init, int32_t *out_ids, void **arena_base,
                       std::vector<size_t> &o_word, std::vector<size_t> &o_val, size_t *total_bytes) {
    size_t total = 0;
    o_word.resize(n_vecs); o_val.resize(n_vecs);
    for (int k = 0; k < n_vecs; k++) {
        o_val[k] = total; total += (((size_t)cap[k] * 8 + 255) / 256) * 256;
        o_word[k] = total; total += (((size_t)cap[k] * 4 + 255) / 256) * 256;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Arena ar;
    { int rc = orbm_arena_alloc(ctx, total, &ar); if (rc) return rc; }
    ar.live = n_vecs;
    int arena_id = -1;
    for (size_t i = 0; i < ctx->bvec_arenas.size(); i++) if (!ctx->bvec_arenas[i].base) { arena_id = (int)i; break; }
    if (arena_id < 0) { ctx->bvec_arenas.push_back(ar); arena_id = (int)ctx->bvec_arenas.size() - 1; } else ctx->bvec_arenas[arena_id] = ar;
    char *b = (char *)ar.base;
    size_t lo = SIZE_MAX, hi = 0;
    for (int k = 0; k < n_vecs; k++) {
        BowVecDev d;
        d.n = n_init[k]; d.word = (const uint32_t *)(b + o_word[k]); d.value = (const double *)(b + o_val[k]);
        int32_t id;
        if (!ctx->free_bvecs.empty()) { id = ctx->free_bvecs.back(); ctx->free_bvecs.pop_back(); ctx->h_bvecs[id] = d; ctx->bvec_arena[id] = arena_id; }
        else { id = (int32_t)ctx->h_bvecs.size(); ctx->h_bvecs.push_back(d); ctx->bvec_arena.push_back(arena_id); }
        out_ids[k] = id;
        lo = std::min(lo, (size_t)id); hi = std::max(hi, (size_t)id);
    }
    if (ctx->h_bvecs.size() > ctx->d_bvecs_cap) {
        size_t ncap = ctx->d_bvecs_cap ? ctx->d_bvecs_cap : 1024;
        while (ncap < ctx->h_bvecs.size()) ncap *= 2;
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        BowVecDev *nt = nullptr;
        HIPCHK(ctx, hipMalloc(&nt, ncap * sizeof(BowVecDev)));
        if (ctx->d_bvecs) HIPCHK(ctx, hipFree(ctx->d_bvecs));
        ctx->d_bvecs = nt; ctx->d_bvecs_cap = ncap;
        lo = 0; hi = ctx->h_bvecs.size() - 1;
    }
    // the table rows travel stream-ordered from a private copy (the vector may grow before the copy has run)
    const size_t tab_bytes = (hi - lo + 1) * sizeof(BowVecDev);
    if (char *pin = (char *)orbm_pin_up(ctx, tab_bytes)) {
        memcpy(pin, ctx->h_bvecs.data() + lo, tab_bytes);
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_bvecs + lo, pin, tab_bytes, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));    // the pinned block is reused by the next upload
    } else {
        HIPCHK(ctx, hipMemcpy(ctx->d_bvecs + lo, ctx->h_bvecs.data() + lo, tab_bytes, hipMemcpyHostToDevice));
    }
    *arena_base = ar.base;
    *total_bytes = total;
    return ORBM_OK;
}

extern "C" int orbm_bowvecs_upload(orbm_ctx *ctx, int n_vecs, const orbm_bowvec_view *views, int32_t *out_ids) {
    if (!ctx) return ORBM_EINVAL;
    if (n_vecs < 0 || (n_vecs > 0 && (!views || !out_ids))) return orbm_fail(ctx, ORBM_EINVAL, "bowvecs_upload: null argument");
    if (n_vecs == 0) return ORBM_OK;
    std::vector<int32_t> ns(n_vecs);
    for (int k = 0; k < n_vecs; k++) {
        const orbm_bowvec_view &v = views[k];
        if (v.n < 0 || v.n > BVEC_MAX_WORDS) return orbm_fail(ctx, ORBM_ELIMIT, "bowvec %d: n = %d outside [0, %d]", k, v.n, BVEC_MAX_WORDS);
        if (v.n > 0 && (!v.word || !v.value)) return orbm_fail(ctx, ORBM_EINVAL, "bowvec %d: null array", k);
        for (int i = 1; i < v.n; i++)
            if (v.word[i] <= v.word[i - 1]) return orbm_fail(ctx, ORBM_EINVAL, "bowvec %d: word ids not strictly ascending at %d", k, i);
        if (v.n > 0 && v.word[v.n - 1] == 0xFFFFFFFFu) return orbm_fail(ctx, ORBM_EINVAL, "bowvec %d: word id 0xFFFFFFFF is reserved", k);
        ns[k] = v.n;
    }
    std::vector<size_t> o_word, o_val;
    void *base = nullptr;
    size_t total = 0;
    int rc = orbm_bvecs_reserve(ctx, n_vecs, ns.data(), ns.data(), out_ids, &base, o_word, o_val, &total);
    if (rc) return rc;
    if (total == 0) return ORBM_OK;
    char *pin = (char *)orbm_pin_up(ctx, total);
    std::vector<uint8_t> pageable;
    if (!pin) { pageable.resize(total); pin = (char *)pageable.data(); }
    for (int k = 0; k < n_vecs; k++) {
        const orbm_bowvec_view &v = views[k];
        if (v.n > 0) { memcpy(pin + o_val[k], v.value, (size_t)v.n * 8); memcpy(pin + o_word[k], v.word, (size_t)v.n * 4); }
    }
    HIPCHK(ctx, hipMemcpyAsync(base, pin, total, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return ORBM_OK;
}

static int check_bvecs(orbm_ctx *ctx, int n, const int32_t *ids, const char *what) {
    for (int i = 0; i < n; i++) {
        const int32_t id = ids[i];
        if (id < 0 || (size_t)id >= ctx->h_bvecs.size() || ctx->bvec_arena[id] < 0)
            return orbm_fail(ctx, ORBM_EINVAL, "%s[%d] = %d is not a live BowVector id", what, i, id);
    }
    return ORBM_OK;
}

extern "C" int orbm_bowvecs_free(orbm_ctx *ctx, int n_vecs, const int32_t *ids) {
    if (!ctx) return ORBM_EINVAL;
    if (n_vecs < 0 || (n_vecs > 0 && !ids)) return orbm_fail(ctx, ORBM_EINVAL, "bowvecs_free: null argument");
    int rc = check_bvecs(ctx, n_vecs, ids, "ids");
    if (rc) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n_vecs; i++) {
        const int32_t id = ids[i];
        const int a = ctx->bvec_arena[id];
        if (a < 0) continue;                               // duplicate in the list
        if (--ctx->bvec_arenas[a].live == 0) orbm_arena_release(ctx, ctx->bvec_arenas[a]);
        ctx->bvec_arena[id] = -1;
        ctx->free_bvecs.push_back(id);
    }
    return ORBM_OK;
}

extern "C" int orbm_bow_score_batch(orbm_ctx *ctx, int n_pairs, const int32_t *vec_a, const int32_t *vec_b, double *score,
                                    int32_t *common, int io_on_device) {
    if (!ctx) return ORBM_EINVAL;
    if (n_pairs < 0 || (n_pairs > 0 && (!vec_a || !vec_b || !score))) return orbm_fail(ctx, ORBM_EINVAL, "bow_score_batch: null argument");
    if (n_pairs == 0) return ORBM_OK;
    int max_a = BVEC_MAX_WORDS;
    if (!io_on_device) {
        int rc = check_bvecs(ctx, n_pairs, vec_a, "vec_a");
        if (rc || (rc = check_bvecs(ctx, n_pairs, vec_b, "vec_b"))) return rc;
        max_a = 1;
        for (int k = 0; k < n_pairs; k++) max_a = std::max(max_a, ctx->h_bvecs[vec_a[k]].n);
    } else if (!ctx->d_bvecs) {
        return orbm_fail(ctx, ORBM_EINVAL, "bow_score_batch: no BowVector has been uploaded");
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = orbm_ws_reserve(ctx, (size_t)n_pairs * 20 + 4096);
    if (rc) return rc;
    IoStage io(ctx, io_on_device);
    const void *da, *db; void *ds, *dc;
    if ((rc = io.in(vec_a, (size_t)n_pairs * 4, &da)) || (rc = io.in(vec_b, (size_t)n_pairs * 4, &db)) ||
        (rc = io.out(score, (size_t)n_pairs * 8, &ds)) || (rc = io.out(common, (size_t)n_pairs * 4, &dc)))
        return rc;
    const size_t lds = (size_t)max_a * 4;
    if (lds > 48 * 1024)
        HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&bow_score_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(bow_score_kernel, dim3((unsigned)n_pairs), dim3(64), lds, ctx->stream, ctx->d_bvecs, (const int32_t *)da,
                       (const int32_t *)db, (double *)ds, (int32_t *)dc);
    HIPCHK(ctx, hipGetLastError());
    return io.finish();
}

extern "C" int orbm_bowvec_download(orbm_ctx *ctx, int32_t id, int32_t *n, uint32_t *word, double *value, int32_t capacity) {
    if (!ctx) return ORBM_EINVAL;
    if (!n) return orbm_fail(ctx, ORBM_EINVAL, "bowvec_download: null argument");
    int rc = check_bvecs(ctx, 1, &id, "id");
    if (rc) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const BowVecDev &h = ctx->h_bvecs[id];
    *n = h.n;
    if (h.n > capacity) return orbm_fail(ctx, ORBM_EINVAL, "bowvec_download: %d words, room for %d", h.n, capacity);
    if (h.n > 0) {
        if (!word || !value) return orbm_fail(ctx, ORBM_EINVAL, "bowvec_download: null array");
        HIPCHK(ctx, hipMemcpy(word, h.word, (size_t)h.n * 4, hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(value, h.value, (size_t)h.n * 8, hipMemcpyDeviceToHost));
    }
    return ORBM_OK;
}
