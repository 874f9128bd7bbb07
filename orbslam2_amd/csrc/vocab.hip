// vocab.hip -- SURVEY 8(f) rank 3: the word / node part of DBoW2's TemplatedVocabulary::transform for ORB descriptors
// (called by Frame::ComputeBoW, Frame.cc:526-533, and KeyFrame::ComputeBoW, KeyFrame.cc:69-78, with levelsup = 4) and
// the FeatureVector it fills, computed and kept on the device: the searches of bow.hip then need no host-built
// FeatureVector upload.  DBoW2 is not in the reference tree; the descent is its published algorithm:
//   final_id = root; do { final_id = the child with the least Hamming distance (first child wins ties);
//                         if (level == L - levelsup) nid = final_id; } while (!leaf);   word = word_id[final_id]
// and FeatureVector::addFeature(nid, i) in feature order == CSR with ascending node ids, ascending indices per node.
//
// transform_kernel: one lane per feature, the children of a node are read from the resident tree (contiguous ids when
// the tree is stored breadth-first, as DBoW2 saves it).  fv_build_kernel: one workgroup per frame, bitonic sort of
// (node id << 32 | feature index) in LDS, run heads -> fv_node / fv_off, indices -> fv_idx, and the frame's n_nodes.
#include <algorithm>
#include <string.h>
#include "orbm_internal.h"

namespace {

constexpr int FV_MAX_N = 8192;     // features per frame the LDS sort holds (64 KB of 8-byte keys)

constexpr int VOC_GROUP = 16;      // lanes per feature: one child each per pass (k = 10 in ORBvoc.txt)
constexpr int VOC_MAX_CHILDREN = 65535;

// NF descriptors down the tree per 16-lane group -> {word id | stopped << 31, node id at level depth - levelsup} each.
// Lane `sub` of a group takes child sub, sub + 16, ...: the 32-byte rows of a node's children are adjacent in memory when
// ids are breadth-first, so a group's loads touch 3 cache lines per level instead of one line per lane and child.
// The least (distance << 16 | child position) over the group is DBoW2's choice: strict '<' over the children in stored
// order = the first child among equals.  A level is two dependent loads (node record, child rows) and the kernel is
// bound by that latency: NF independent descents per group keep NF times as many loads in flight.  Control flow is
// wave-uniform (descents that reached a leaf idle until the whole wave has), loads of lanes without a child go to
// row 0, so the cross-lane steps never read a lane that sits out.
template <int NF>
__device__ __forceinline__ void descend(const VocabDev &V, const gptr<const v4u> (&g)[NF], int levelsup, int32_t (&word)[NF], int (&nid)[NF], int (&leaf)[NF]) {
    const uint32_t sub = threadIdx.x & (VOC_GROUP - 1);
    uint32_t q[NF][8];
#pragma unroll
    for (int u = 0; u < NF; u++) {
        const v4u lo = g[u][0], hi = g[u][1];
        q[u][0] = lo.x; q[u][1] = lo.y; q[u][2] = lo.z; q[u][3] = lo.w; q[u][4] = hi.x; q[u][5] = hi.y; q[u][6] = hi.z; q[u][7] = hi.w;
    }
    const int nid_level = V.depth - levelsup;
    uint32_t final_id[NF];
    int level[NF];
#pragma unroll
    for (int u = 0; u < NF; u++) { nid[u] = 0; final_id[u] = 0; level[u] = 0; }       // nid_level <= 0: the root
    for (;;) {
        uint2 rec[NF];
        uint32_t cnt[NF], maxcnt = 0;
        bool all_contiguous = true;
#pragma unroll
        for (int u = 0; u < NF; u++) {
            rec[u] = V.node_rec[final_id[u]];
            cnt[u] = rec[u].y & 0x7FFFFFFFu;
            maxcnt = max(maxcnt, cnt[u]);
            all_contiguous = all_contiguous && (cnt[u] == 0 || (rec[u].y >> 31) != 0);
        }
        if (__all(maxcnt == 0)) break;             // every descent of the wave is at a leaf
        const bool wave_contiguous = __all(all_contiguous);
        uint32_t best[NF];
#pragma unroll
        for (int u = 0; u < NF; u++) best[u] = 0xFFFFFFFFu;
        for (uint32_t c0 = 0; __any(c0 < maxcnt); c0 += VOC_GROUP) {
            const uint32_t j = c0 + sub;
            uint32_t id[NF];
#pragma unroll
            for (int u = 0; u < NF; u++) id[u] = j < cnt[u] ? rec[u].x + j : 0u;
            if (!wave_contiguous) {
#pragma unroll
                for (int u = 0; u < NF; u++)
                    if (j < cnt[u] && (rec[u].y >> 31) == 0) id[u] = V.child_idx[id[u]];
            }
            v4u clo[NF], chi[NF];
#pragma unroll
            for (int u = 0; u < NF; u++) {
                const gptr<const v4u> cg = (gptr<const v4u>)V.desc + 2 * (size_t)id[u];
                clo[u] = cg[0]; chi[u] = cg[1];
            }
#pragma unroll
            for (int u = 0; u < NF; u++) {
                const uint32_t cd[8] = {clo[u].x, clo[u].y, clo[u].z, clo[u].w, chi[u].x, chi[u].y, chi[u].z, chi[u].w};
                const uint32_t key = (hamming256(q[u], cd) << 16) | j;
                best[u] = min(best[u], j < cnt[u] ? key : 0xFFFFFFFFu);
            }
        }
#pragma unroll
        for (int m = 1; m < VOC_GROUP; m <<= 1) {
#pragma unroll
            for (int u = 0; u < NF; u++) best[u] = min(best[u], (uint32_t)__shfl_xor((int)best[u], m, VOC_GROUP));
        }
#pragma unroll
        for (int u = 0; u < NF; u++)
            if (cnt[u] != 0) {
                const uint32_t j = best[u] & 0xFFFFu;
                final_id[u] = (rec[u].y >> 31) ? rec[u].x + j : V.child_idx[rec[u].x + j];
                ++level[u];
                if (level[u] == nid_level) nid[u] = (int)final_id[u];
            }
    }
#pragma unroll
    for (int u = 0; u < NF; u++) { word[u] = V.word_id[final_id[u]]; leaf[u] = (int)final_id[u]; }       // bit 31 of word: the word's weight is not positive
}

template <int NF>
__global__ __launch_bounds__(256) void transform_kernel(const FrameDev *__restrict__ table, const int32_t *__restrict__ ids,
                                                        VocabDev V, int levelsup, int blocks_per_frame, int stride,
                                                        int32_t *__restrict__ word_out, int32_t *__restrict__ node_out,
                                                        int32_t *__restrict__ leaf_out) {
    const int f = blockIdx.x / blocks_per_frame;
    const int i0 = ((blockIdx.x - f * blocks_per_frame) * (256 / VOC_GROUP) + (int)(threadIdx.x / VOC_GROUP)) * NF;
    const FrameG F(table[ids[f]]);
    const int n = F.n;
    if (n == 0) return;                            // block-uniform
    gptr<const v4u> g[NF];
#pragma unroll
    for (int u = 0; u < NF; u++) g[u] = (gptr<const v4u>)F.desc + 2 * (size_t)min(i0 + u, n - 1);
    int32_t w[NF]; int nid[NF], leaf[NF];
    descend<NF>(V, g, levelsup, w, nid, leaf);
    const int sub = (int)(threadIdx.x & (VOC_GROUP - 1));
#pragma unroll
    for (int u = 0; u < NF; u++)
        if (sub == u && i0 + u < n) {
            word_out[(size_t)f * stride + i0 + u] = w[u] & 0x7FFFFFFF;
            node_out[(size_t)f * stride + i0 + u] = w[u] < 0 ? -1 : nid[u];
            if (leaf_out) leaf_out[(size_t)f * stride + i0 + u] = w[u] < 0 ? -1 : leaf[u];
        }
}

template <int NF>
__global__ __launch_bounds__(256) void transform_flat_kernel(const uint32_t *__restrict__ desc, int n, VocabDev V, int levelsup,
                                                             int32_t *__restrict__ word_out, int32_t *__restrict__ node_out) {
    const int i0 = (blockIdx.x * (256 / VOC_GROUP) + (int)(threadIdx.x / VOC_GROUP)) * NF;
    gptr<const v4u> g[NF];
#pragma unroll
    for (int u = 0; u < NF; u++) g[u] = (gptr<const v4u>)desc + 2 * (size_t)min(i0 + u, n - 1);
    int32_t w[NF]; int nid[NF], leaf[NF];
    descend<NF>(V, g, levelsup, w, nid, leaf);
    const int sub = (int)(threadIdx.x & (VOC_GROUP - 1));
#pragma unroll
    for (int u = 0; u < NF; u++)
        if (sub == u && i0 + u < n) {
            word_out[i0 + u] = w[u] & 0x7FFFFFFF;
            node_out[i0 + u] = w[u] < 0 ? -1 : nid[u];
        }
}

// descents per 16-lane group: 4 once there is enough work to fill the chip either way
int g_vocab_variant = 0;      // test knob: 0 = by size, 1 / 4 = that many descents per group
inline int descents_per_group(size_t total_features) {
    if (g_vocab_variant == 1 || g_vocab_variant == 4) return g_vocab_variant;
    return total_features >= (size_t)256 * 1024 ? 4 : 1;
}

constexpr int FVB_THREADS = 1024;

__global__ __launch_bounds__(FVB_THREADS) void fv_build_kernel(FrameDev *__restrict__ table, const int32_t *__restrict__ ids, int stride,
                                                               const int32_t *__restrict__ node_in, int32_t *__restrict__ n_nodes_out) {
    extern __shared__ __align__(16) unsigned long long keys[];
    __shared__ uint32_t part[FVB_THREADS];
    __shared__ int n_in;
    const int f = blockIdx.x, tid = threadIdx.x;
    FrameDev &F = table[ids[f]];
    const int n_feat = F.n;
    int m = 1;
    while (m < n_feat) m <<= 1;
    if (tid == 0) n_in = 0;
    __syncthreads();
    int mine = 0;
    for (int i = tid; i < m; i += FVB_THREADS) {
        const int node = i < n_feat ? node_in[(size_t)f * stride + i] : -1;
        keys[i] = node >= 0 ? ((unsigned long long)(uint32_t)node << 32) | (uint32_t)i : ~0ull;     // left out: sorts last
        mine += node >= 0;
    }
    if (mine) atomicAdd(&n_in, mine);
    __syncthreads();
    const int n = n_in;
    // bitonic sort, one compare-exchange per thread and pass over m / 2 pairs
    for (int k = 2; k <= m; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (m >> 1); t += FVB_THREADS) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));     // the pair's lower index: bit j clear
                const int l = i | j;
                const unsigned long long a = keys[i], b = keys[l];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { keys[i] = b; keys[l] = a; }
            }
            __syncthreads();
        }
    const gptr<uint32_t> fv_node = (gptr<uint32_t>)F.fv_node, fv_off = (gptr<uint32_t>)F.fv_off, fv_idx = (gptr<uint32_t>)F.fv_idx;
    // run heads: thread t owns positions [t*per, (t+1)*per); exclusive scan of the head counts over the workgroup
    const int per = (n + FVB_THREADS - 1) / FVB_THREADS;
    const int p0 = min(n, tid * per), p1 = min(n, p0 + per);
    uint32_t heads = 0;
    for (int p = p0; p < p1; p++) heads += (p == 0 || (keys[p] >> 32) != (keys[p - 1] >> 32)) ? 1u : 0u;
    part[tid] = heads;
    __syncthreads();
    for (int d = 1; d < FVB_THREADS; d <<= 1) {
        const uint32_t add = tid >= d ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    uint32_t base = part[tid] - heads;
    const uint32_t total = part[FVB_THREADS - 1];
    for (int p = p0; p < p1; p++) {
        const unsigned long long kp = keys[p];
        fv_idx[p] = (uint32_t)kp;
        if (p == 0 || (kp >> 32) != (keys[p - 1] >> 32)) { fv_node[base] = (uint32_t)(kp >> 32); fv_off[base] = (uint32_t)p; base++; }
    }
    // the FeatureVector-ordered operand image (descfv), if one was built, describes the old order: dropped (the host side
    // does the same to its mirror and releases the memory, orbm_drop_fv_image)
    if (tid == 0) { fv_off[total] = (uint32_t)n; F.n_nodes = (int32_t)total; F.descfv = nullptr; n_nodes_out[f] = (int32_t)total; }
}

// BowVector of each frame (TemplatedVocabulary::transform, BowVector part): words ascending; value = the weights of the
// word's features added one by one in feature order (addWeight), or the first one only (addIfNotExist, IDF / BINARY
// weighting); then divided by the L1 norm, itself a sum over the words in ascending order (BowVector::normalize), or by
// the number of words when the scoring does not normalise.  Both sums are sequential in DBoW2 and so they are here:
// one thread per word adds its features' weights, thread 0 adds the norm.  leaf_in = leaf node per feature, -1 = left out.
__global__ __launch_bounds__(FVB_THREADS) void bv_build_kernel(const FrameDev *__restrict__ table, const int32_t *__restrict__ ids, int stride,
                                                               const int32_t *__restrict__ word_in, const int32_t *__restrict__ leaf_in,
                                                               VocabDev V, BowVecDev *__restrict__ bv_table, const int32_t *__restrict__ bv_ids,
                                                               int32_t *__restrict__ n_words_out) {
    extern __shared__ __align__(16) unsigned long long keys[];       // m keys, then m doubles
    __shared__ uint32_t part[FVB_THREADS];
    __shared__ int n_in;
    __shared__ double norm_sh;
    const int f = blockIdx.x, tid = threadIdx.x;
    const int n_feat = table[ids[f]].n;
    int m = 1;
    while (m < n_feat) m <<= 1;
    double *sums = reinterpret_cast<double *>(keys + max(m, 2));
    if (tid == 0) n_in = 0;
    __syncthreads();
    int mine = 0;
    for (int i = tid; i < m; i += FVB_THREADS) {
        const bool in = i < n_feat && leaf_in[(size_t)f * stride + i] >= 0;
        keys[i] = in ? ((unsigned long long)(uint32_t)word_in[(size_t)f * stride + i] << 32) | (uint32_t)i : ~0ull;
        mine += in;
    }
    if (mine) atomicAdd(&n_in, mine);
    __syncthreads();
    const int n = n_in;
    for (int k = 2; k <= m; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (m >> 1); t += FVB_THREADS) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const unsigned long long a = keys[i], b = keys[l];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { keys[i] = b; keys[l] = a; }
            }
            __syncthreads();
        }
    BowVecDev &out = bv_table[bv_ids[f]];
    const gptr<uint32_t> o_word = (gptr<uint32_t>)out.word;
    const gptr<double> o_val = (gptr<double>)out.value;
    const int per = (n + FVB_THREADS - 1) / FVB_THREADS;
    const int p0 = min(n, tid * per), p1 = min(n, p0 + per);
    uint32_t heads = 0;
    for (int p = p0; p < p1; p++) heads += (p == 0 || (keys[p] >> 32) != (keys[p - 1] >> 32)) ? 1u : 0u;
    part[tid] = heads;
    __syncthreads();
    for (int d = 1; d < FVB_THREADS; d <<= 1) {
        const uint32_t add = tid >= d ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    uint32_t base = part[tid] - heads;
    const int n_words = (int)part[FVB_THREADS - 1];
    const bool accumulate = V.weighting == 0 || V.weighting == 1;          // TF_IDF, TF
    for (int p = p0; p < p1; p++) {
        const unsigned long long kp = keys[p];
        if (p == 0 || (kp >> 32) != (keys[p - 1] >> 32)) {
            // run head: this thread owns the whole run (it may extend into the next thread's positions)
            double acc = V.weight[leaf_in[(size_t)f * stride + (uint32_t)kp]];
            if (accumulate)
                for (int r = p + 1; r < n && (keys[r] >> 32) == (kp >> 32); r++) acc += V.weight[leaf_in[(size_t)f * stride + (uint32_t)keys[r]]];
            o_word[base] = (uint32_t)(kp >> 32);
            sums[base] = acc;
            base++;
        }
    }
    __syncthreads();
    const bool l1 = V.scoring != 5;                                        // everything but DOT_PRODUCT (L2_NORM is refused by the host)
    if (tid == 0) {
        double norm = 0.0;
        if (l1) for (int w = 0; w < n_words; w++) norm += fabs(sums[w]);
        else norm = (double)n_words;
        norm_sh = norm;
        out.n = n_words;
        n_words_out[f] = n_words;
    }
    __syncthreads();
    const double norm = norm_sh;
    // normalize(): only if the norm is positive; the unnormalised TF case divides by the word count only when accumulating
    const bool divide = l1 ? norm > 0.0 : (accumulate && n_words > 0);
    for (int w = tid; w < n_words; w += FVB_THREADS) o_val[w] = divide ? sums[w] / norm : sums[w];
}

}  // namespace

extern "C" void orbm__vocab_variant(int v) { g_vocab_variant = v; }

extern "C" int orbm_vocab_upload(orbm_ctx *ctx, const orbm_vocab_view *v, int32_t *vocab_id) {
    if (!ctx) return ORBM_EINVAL;
    if (!v || !vocab_id) return orbm_fail(ctx, ORBM_EINVAL, "vocab_upload: null argument");
    if (v->n_nodes < 2 || !v->child_off || !v->desc || !v->word_id)
        return orbm_fail(ctx, ORBM_EINVAL, "vocab_upload: empty tree or null array");
    const size_t nn = (size_t)v->n_nodes;
    const uint32_t nchild = v->child_off[nn];
    if (v->weighting < 0 || v->weighting > 3 || v->scoring < 0 || v->scoring > 5)
        return orbm_fail(ctx, ORBM_EINVAL, "vocab_upload: weighting %d / scoring %d outside DBoW2's enums", v->weighting, v->scoring);
    if (v->child_off[0] != 0 || v->child_off[1] == 0 || !v->child_idx) return orbm_fail(ctx, ORBM_EINVAL, "vocab_upload: bad child_off or a root without children");
    // every node but the root is the child of exactly one EARLIER node: the descent then always terminates
    std::vector<uint8_t> seen(nn, 0);
    for (size_t i = 0; i < nn; i++) {
        if (v->child_off[i + 1] < v->child_off[i]) return orbm_fail(ctx, ORBM_EINVAL, "vocab_upload: child_off not monotone at node %zu", i);
        for (uint32_t c = v->child_off[i]; c < v->child_off[i + 1]; c++) {
            const uint32_t id = v->child_idx[c];
            if (id >= nn || id <= i || seen[id]) return orbm_fail(ctx, ORBM_EINVAL, "vocab_upload: node %zu has a bad child %u", i, id);
            seen[id] = 1;
        }
        if (v->child_off[i + 1] == v->child_off[i] && v->word_id[i] < 0)
            return orbm_fail(ctx, ORBM_EINVAL, "vocab_upload: leaf %zu has no word id", i);
        if (v->child_off[i + 1] - v->child_off[i] > (uint32_t)VOC_MAX_CHILDREN)
            return orbm_fail(ctx, ORBM_ELIMIT, "vocab_upload: node %zu has more than %d children", i, VOC_MAX_CHILDREN);
    }
    std::vector<uint2> rec(nn);
    for (size_t i = 0; i < nn; i++) {
        const uint32_t b = v->child_off[i], cnt = v->child_off[i + 1] - b;
        bool contiguous = cnt > 0;
        for (uint32_t c = 1; c < cnt && contiguous; c++) contiguous = v->child_idx[b + c] == v->child_idx[b] + c;
        rec[i] = contiguous ? make_uint2(v->child_idx[b], cnt | 0x80000000u) : make_uint2(b, cnt);
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t o_off = 0, o_idx = nn * 8, o_desc = ((o_idx + (size_t)nchild * 4 + 255) / 256) * 256;
    const size_t o_word = o_desc + nn * 32, o_weight = ((o_word + nn * 4 + 255) / 256) * 256, total = o_weight + nn * 8;
    VocabDev V;
    { int rc = orbm_arena_alloc(ctx, total, &V.arena); if (rc) return rc; }
    char *b = (char *)V.arena.base;
    HIPCHK(ctx, hipMemcpy(b + o_off, rec.data(), nn * 8, hipMemcpyHostToDevice));
    if (nchild) HIPCHK(ctx, hipMemcpy(b + o_idx, v->child_idx, (size_t)nchild * 4, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(b + o_desc, v->desc, nn * 32, hipMemcpyHostToDevice));
    std::vector<int32_t> words(v->word_id, v->word_id + nn);
    for (size_t i = 0; i < nn; i++) {
        const bool leaf = v->child_off[i + 1] == v->child_off[i];
        if (leaf && v->weight && !(v->weight[i] > 0)) words[i] |= (int32_t)0x80000000;     // "if (w > 0)" of transform()
    }
    HIPCHK(ctx, hipMemcpy(b + o_word, words.data(), nn * 4, hipMemcpyHostToDevice));
    {
        std::vector<double> wts(nn, 1.0);
        if (v->weight) wts.assign(v->weight, v->weight + nn);
        HIPCHK(ctx, hipMemcpy(b + o_weight, wts.data(), nn * 8, hipMemcpyHostToDevice));
    }
    V.weight = (const double *)(b + o_weight);
    V.weighting = v->weighting; V.scoring = v->scoring;
    V.n_nodes = v->n_nodes; V.depth = v->depth;
    V.node_rec = (const uint2 *)(b + o_off); V.child_idx = (const uint32_t *)(b + o_idx);
    V.desc = (const uint32_t *)(b + o_desc); V.word_id = (const int32_t *)(b + o_word);
    int id = -1;
    for (size_t i = 0; i < ctx->vocabs.size(); i++) if (!ctx->vocabs[i].arena.base) { id = (int)i; break; }
    if (id < 0) { ctx->vocabs.push_back(V); id = (int)ctx->vocabs.size() - 1; } else ctx->vocabs[id] = V;
    *vocab_id = id;
    return ORBM_OK;
}

extern "C" int orbm_vocab_free(orbm_ctx *ctx, int32_t vocab_id) {
    if (!ctx) return ORBM_EINVAL;
    if (vocab_id < 0 || (size_t)vocab_id >= ctx->vocabs.size() || !ctx->vocabs[vocab_id].arena.base)
        return orbm_fail(ctx, ORBM_EINVAL, "vocab_free: %d is not a live vocabulary", vocab_id);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    orbm_arena_release(ctx, ctx->vocabs[vocab_id].arena);
    ctx->vocabs[vocab_id] = VocabDev();
    return ORBM_OK;
}

extern "C" int orbm_frames_compute_bow(orbm_ctx *ctx, int32_t vocab_id, int n_frames, const int32_t *frames, int levelsup,
                                       int32_t stride, int32_t *word_id, int32_t *node_id, int32_t *bowvec_ids) {
    if (!ctx) return ORBM_EINVAL;
    if (vocab_id < 0 || (size_t)vocab_id >= ctx->vocabs.size() || !ctx->vocabs[vocab_id].arena.base)
        return orbm_fail(ctx, ORBM_EINVAL, "frames_compute_bow: %d is not a live vocabulary", vocab_id);
    if (n_frames < 0 || (n_frames > 0 && !frames) || stride < 0) return orbm_fail(ctx, ORBM_EINVAL, "frames_compute_bow: null argument or negative size");
    if (n_frames == 0) return ORBM_OK;
    int rc = orbm_check_frames(ctx, n_frames, frames, "frames");
    if (rc) return rc;
    int max_n = 0;
    for (int k = 0; k < n_frames; k++) {
        for (int j = 0; j < k; j++) if (frames[j] == frames[k]) return orbm_fail(ctx, ORBM_EINVAL, "frames_compute_bow: frame %d listed twice", frames[k]);
        max_n = std::max(max_n, ctx->h_frames[frames[k]].n);
    }
    if (max_n > FV_MAX_N) return orbm_fail(ctx, ORBM_ELIMIT, "frames_compute_bow: frames of more than %d features are not supported (got %d)", FV_MAX_N, max_n);
    if ((word_id || node_id) && max_n > stride) return orbm_fail(ctx, ORBM_EINVAL, "frames_compute_bow: stride %d < largest frame %d", stride, max_n);
    if (bowvec_ids && ctx->vocabs[vocab_id].scoring == 1)
        return orbm_fail(ctx, ORBM_EINVAL, "frames_compute_bow: BowVectors of an L2_NORM vocabulary are not built on the device");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int st = std::max(std::max(stride, max_n), 1);
    const size_t elems = (size_t)n_frames * st;
    rc = orbm_ws_reserve(ctx, (size_t)n_frames * 16 + (word_id ? elems * 4 : 0) + (node_id ? elems * 4 : 0) + 4096);
    if (rc) return rc;
    const size_t col = ((elems * 4 + 255) / 256) * 256;
    rc = orbm_ws2_reserve(ctx, col * 3 + 512);           // outputs the caller does not ask for stay on the device
    if (rc) return rc;
    // room for the BowVectors (may grow and re-send the BowVector table)
    std::vector<size_t> bo_word, bo_val;
    if (bowvec_ids) {
        std::vector<int32_t> caps(n_frames), zeros(n_frames, 0);
        for (int k = 0; k < n_frames; k++) caps[k] = std::max(ctx->h_frames[frames[k]].n, 1);
        void *base; size_t total;
        rc = orbm_bvecs_reserve(ctx, n_frames, caps.data(), zeros.data(), bowvec_ids, &base, bo_word, bo_val, &total);
        if (rc) return rc;
    }
    IoStage io(ctx, 0);
    const void *dids, *dbv = nullptr; void *dw, *dn, *dnn, *dnw = nullptr;
    std::vector<int32_t> nn(n_frames), nw(n_frames);
    if ((rc = io.in(frames, (size_t)n_frames * 4, &dids)) || (bowvec_ids && (rc = io.in(bowvec_ids, (size_t)n_frames * 4, &dbv))) ||
        (rc = io.out(nn.data(), (size_t)n_frames * 4, &dnn)) || (bowvec_ids && (rc = io.out(nw.data(), (size_t)n_frames * 4, &dnw))) ||
        (rc = io.out(word_id, elems * 4, &dw)) || (rc = io.out(node_id, elems * 4, &dn)))
        return rc;
    if (!dw) dw = ctx->ws2;
    if (!dn) dn = (char *)ctx->ws2 + col;
    int32_t *dleaf = bowvec_ids ? (int32_t *)((char *)ctx->ws2 + 2 * col) : nullptr;
    HIPCHK(ctx, hipMemsetAsync(dw, 0xFF, elems * 4, ctx->stream));         // rows beyond a frame's n stay -1
    HIPCHK(ctx, hipMemsetAsync(dn, 0xFF, elems * 4, ctx->stream));
    const VocabDev &V = ctx->vocabs[vocab_id];
    if (max_n > 0) {
        const int nf = descents_per_group((size_t)n_frames * max_n);
        const int per_block = 256 / VOC_GROUP * nf, bpf = (max_n + per_block - 1) / per_block;
        if (nf == 4)
            hipLaunchKernelGGL(transform_kernel<4>, dim3((unsigned)(n_frames * bpf)), dim3(256), 0, ctx->stream, ctx->d_frames,
                               (const int32_t *)dids, V, levelsup, bpf, st, (int32_t *)dw, (int32_t *)dn, dleaf);
        else
            hipLaunchKernelGGL(transform_kernel<1>, dim3((unsigned)(n_frames * bpf)), dim3(256), 0, ctx->stream, ctx->d_frames,
                               (const int32_t *)dids, V, levelsup, bpf, st, (int32_t *)dw, (int32_t *)dn, dleaf);
        HIPCHK(ctx, hipGetLastError());
    }
    int m = 1;
    while (m < max_n) m <<= 1;
    const size_t lds = (size_t)std::max(m, 2) * 8;
    if (lds > 48 * 1024)
        HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&fv_build_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(fv_build_kernel, dim3(n_frames), dim3(FVB_THREADS), lds, ctx->stream, ctx->d_frames, (const int32_t *)dids, st,
                       (const int32_t *)dn, (int32_t *)dnn);
    HIPCHK(ctx, hipGetLastError());
    if (bowvec_ids) {
        const size_t lds2 = (size_t)std::max(m, 2) * 16;
        if (lds2 > 48 * 1024)
            HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&bv_build_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        hipLaunchKernelGGL(bv_build_kernel, dim3(n_frames), dim3(FVB_THREADS), lds2, ctx->stream, ctx->d_frames, (const int32_t *)dids, st,
                           (const int32_t *)dw, (const int32_t *)dleaf, V, ctx->d_bvecs, (const int32_t *)dbv, (int32_t *)dnw);
        HIPCHK(ctx, hipGetLastError());
    }
    rc = io.finish();
    if (rc) return rc;
    // keep the host mirrors of the tables in step (they are re-sent when a table grows)
    for (int k = 0; k < n_frames; k++) { ctx->h_frames[frames[k]].n_nodes = nn[k]; orbm_drop_fv_image(ctx, frames[k]); }
    if (bowvec_ids) for (int k = 0; k < n_frames; k++) ctx->h_bvecs[bowvec_ids[k]].n = nw[k];
    return ORBM_OK;
}

extern "C" int orbm_frame_fv_download(orbm_ctx *ctx, int32_t frame, int32_t *n_nodes, uint32_t *fv_node, uint32_t *fv_off, uint32_t *fv_idx) {
    if (!ctx) return ORBM_EINVAL;
    if (!n_nodes || !fv_node || !fv_off || !fv_idx) return orbm_fail(ctx, ORBM_EINVAL, "frame_fv_download: null argument");
    int rc = orbm_check_frames(ctx, 1, &frame, "frame");
    if (rc) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const FrameDev &h = ctx->h_frames[frame];
    const int nn = h.n_nodes;
    *n_nodes = nn;
    HIPCHK(ctx, hipMemcpy(fv_off, h.fv_off, ((size_t)nn + 1) * 4, hipMemcpyDeviceToHost));
    if (nn > 0) {
        HIPCHK(ctx, hipMemcpy(fv_node, h.fv_node, (size_t)nn * 4, hipMemcpyDeviceToHost));
        if (fv_off[nn] > (uint32_t)std::max(h.n, 0)) return orbm_fail(ctx, ORBM_EINVAL, "frame_fv_download: FeatureVector of %u entries exceeds the frame's %d features", fv_off[nn], h.n);
        if (fv_off[nn]) HIPCHK(ctx, hipMemcpy(fv_idx, h.fv_idx, (size_t)fv_off[nn] * 4, hipMemcpyDeviceToHost));
    }
    return ORBM_OK;
}

extern "C" int orbm_vocab_transform(orbm_ctx *ctx, int32_t vocab_id, int n, const uint8_t *desc, int levelsup, int32_t *word_id,
                                    int32_t *node_id, int io_on_device) {
    if (!ctx) return ORBM_EINVAL;
    if (vocab_id < 0 || (size_t)vocab_id >= ctx->vocabs.size() || !ctx->vocabs[vocab_id].arena.base)
        return orbm_fail(ctx, ORBM_EINVAL, "vocab_transform: %d is not a live vocabulary", vocab_id);
    if (n < 0 || (n > 0 && (!desc || !word_id || !node_id))) return orbm_fail(ctx, ORBM_EINVAL, "vocab_transform: null argument or negative size");
    if (n == 0) return ORBM_OK;
    if (io_on_device && ((uintptr_t)desc & 15)) return orbm_fail(ctx, ORBM_EINVAL, "vocab_transform: device descriptors must be 16-byte aligned");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = orbm_ws_reserve(ctx, (size_t)n * 40 + 4096);
    if (rc) return rc;
    IoStage io(ctx, io_on_device);
    const void *dd; void *dw, *dn;
    if ((rc = io.in(desc, (size_t)n * 32, &dd)) || (rc = io.out(word_id, (size_t)n * 4, &dw)) || (rc = io.out(node_id, (size_t)n * 4, &dn)))
        return rc;
    const int nf = descents_per_group((size_t)n);
    const int per_block = 256 / VOC_GROUP * nf;
    if (nf == 4)
        hipLaunchKernelGGL(transform_flat_kernel<4>, dim3((unsigned)((n + per_block - 1) / per_block)), dim3(256), 0, ctx->stream, (const uint32_t *)dd, n,
                           ctx->vocabs[vocab_id], levelsup, (int32_t *)dw, (int32_t *)dn);
    else
        hipLaunchKernelGGL(transform_flat_kernel<1>, dim3((unsigned)((n + per_block - 1) / per_block)), dim3(256), 0, ctx->stream, (const uint32_t *)dd, n,
                           ctx->vocabs[vocab_id], levelsup, (int32_t *)dw, (int32_t *)dn);
    HIPCHK(ctx, hipGetLastError());
    return io.finish();
}
