// allpairs.hip -- BASELINE config 5: batched keyframe-pair matching as ONE call that leaves compact match records on
// the device, plus the block-cyclic sharding of the pair triangle over the GPUs of a node.
//
// Reference loops this replaces: LoopClosing.cc:433-477 (SearchByBoW against every consistent candidate, keep a pair only
// if nmatches >= 20, :459) and Tracking.cc:1778-1801 (the same against every relocalisation candidate, >= 15, :1787).
// The match rows (stride x int32 per pair) never leave the GPU: what travels to the caller / to rank 0 is
//   * one 8-byte summary per pair  {u32 pair_id, u16 nmatches, u16 0}
//   * (u16 idx1, u16 idx2) x nmatches, ascending idx1, ONLY for pairs with nmatches >= threshold, pair after pair in
//     batch order (the receiver finds a pair's list by a running sum over the summaries).
#include <algorithm>
#include <string.h>
#include <thread>
#include "orbm_internal.h"

namespace {

constexpr int SCAN_THREADS = 1024;

// summaries + exclusive scan of the accepted pairs' counts (single workgroup: a batch is at most a few 10^5 pairs)
__global__ __launch_bounds__(SCAN_THREADS) void pack_scan_kernel(int n_pairs, const uint32_t *__restrict__ pair_id,
                                                                 const int32_t *__restrict__ counts, int threshold,
                                                                 uint2 *__restrict__ summaries, uint32_t *__restrict__ list_off,
                                                                 uint32_t *__restrict__ totals) {
    __shared__ uint32_t part[SCAN_THREADS / 64];
    __shared__ uint32_t part_acc[SCAN_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t base = 0, accepted = 0;
    for (int p0 = 0; p0 < n_pairs; p0 += SCAN_THREADS) {
        const int p = p0 + tid;
        const uint32_t id = p < n_pairs ? (pair_id ? pair_id[p] : (uint32_t)p) : ORBM_PAIR_PAD;
        const int c = id != ORBM_PAIR_PAD ? counts[p] : 0;                       // padding entries: n = 0, no list
        const bool acc = c >= threshold && c > 0;
        const uint32_t v = acc ? (uint32_t)c : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        const unsigned long long am = __ballot(acc);
        if (lane == 63) part[wv] = incl;
        if (lane == 0) part_acc[wv] = (uint32_t)__popcll(am);
        __syncthreads();
        uint32_t before = 0, all = 0, nacc = 0;
#pragma unroll
        for (int w = 0; w < SCAN_THREADS / 64; w++) {
            const uint32_t s = part[w];
            if (w < wv) before += s;
            all += s;
            nacc += part_acc[w];
        }
        if (p < n_pairs) {
            list_off[p] = acc ? base + before + incl - v : 0xFFFFFFFFu;
            summaries[p] = make_uint2(id, (uint32_t)min(max(c, 0), 65535));
        }
        base += all;
        accepted += nacc;
        __syncthreads();
    }
    if (tid == 0) { totals[0] = accepted; totals[1] = base; }
}

// one wave per accepted pair: the non-negative entries of its match row, in ascending row index, as (row | value << 16)
__global__ __launch_bounds__(256) void pack_lists_kernel(int n_pairs, const int32_t *__restrict__ counts,
                                                         const int32_t *__restrict__ match, int stride,
                                                         const uint32_t *__restrict__ list_off, uint32_t *__restrict__ lists,
                                                         uint32_t list_cap, const FrameDev *__restrict__ table,
                                                         const int32_t *__restrict__ row_frame) {
    const int lane = threadIdx.x & 63;
    const int p = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (p >= n_pairs) return;
    const uint32_t off = list_off[p];
    if (off == 0xFFFFFFFFu) return;                                              // not accepted (or padding)
    const int c = counts[p];
    const gptr<const int32_t> row = in_global(match) + (size_t)p * stride;
    // the searches write rows [0, n) of the frame that indexes the row; beyond that the buffer is only defined when the caller says so
    const int len = table ? min(stride, table[row_frame[p]].n) : stride;
    uint32_t pos = 0;
    for (int i0 = 0; i0 < len && pos < (uint32_t)c; i0 += 64) {
        const int i = i0 + lane;
        const int m = i < len ? row[i] : -1;
        const bool ok = m >= 0;
        const unsigned long long mask = __ballot(ok);
        const uint32_t k = pos + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (ok && k < (uint32_t)c && off + k < list_cap) lists[off + k] = (uint32_t)i | ((uint32_t)m << 16);
        pos += (uint32_t)__popcll(mask);
    }
}

}  // namespace

extern "C" {

int64_t orbm_allpairs_shard(int n_kf, int tile, int rank, int world, int64_t first, int64_t max_pairs, int32_t *pair_a,
                            int32_t *pair_b) {
    if (n_kf < 0 || tile <= 0 || world <= 0 || rank < 0 || rank >= world || first < 0 || max_pairs < 0) return ORBM_EINVAL;
    const int nt = (n_kf + tile - 1) / tile;
    int64_t seen = 0, k = 0;
    const bool fill = pair_a && pair_b && max_pairs > 0;
    for (int ti = 0; ti < nt; ti++)
        for (int tj = ti; tj < nt; tj++, k++) {
            if (k % world != rank) continue;
            const int a0 = ti * tile, a1 = std::min(n_kf, a0 + tile), b0 = tj * tile, b1 = std::min(n_kf, b0 + tile);
            const int64_t in_tile = ti == tj ? (int64_t)(a1 - a0) * (a1 - a0 - 1) / 2 : (int64_t)(a1 - a0) * (b1 - b0);
            if (fill && seen + in_tile > first && seen < first + max_pairs) {
                int64_t q = seen;
                for (int a = a0; a < a1; a++)
                    for (int b = (ti == tj ? a + 1 : b0); b < b1; b++, q++)
                        if (q >= first && q < first + max_pairs) { pair_a[q - first] = a; pair_b[q - first] = b; }
            }
            seen += in_tile;
        }
    return seen;
}

int orbm_pack_records(orbm_ctx *ctx, int n_pairs, const uint32_t *pair_id, const int32_t *counts, const int32_t *match,
                      int32_t stride, int threshold, orbm_pair_summary *summaries, uint32_t *lists, uint32_t list_cap,
                      uint32_t *totals, int io_on_device) {
    if (!ctx) return ORBM_EINVAL;
    if (n_pairs < 0 || stride < 0 || (n_pairs > 0 && (!counts || !match || !summaries)) || !totals || (list_cap > 0 && !lists))
        return orbm_fail(ctx, ORBM_EINVAL, "pack_records: null argument or negative size");
    if (stride > 65536) return orbm_fail(ctx, ORBM_ELIMIT, "pack_records: stride %d: row and value indices must fit 16 bits", stride);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t P = (size_t)n_pairs, rows = P * (size_t)stride;
    int rc = orbm_ws2_reserve(ctx, P * 4 + 256);
    if (rc) return rc;
    IoStage io(ctx, io_on_device);
    if (!io_on_device) {
        rc = orbm_ws_reserve(ctx, P * 16 + rows * 4 + (size_t)list_cap * 4 + 4096);
        if (rc) return rc;
    }
    const void *dpid, *dcnt, *dmatch; void *dsum, *dlists, *dtot;
    if ((rc = io.in(pair_id, P * 4, &dpid)) || (rc = io.in(counts, P * 4, &dcnt)) || (rc = io.in(match, rows * 4, &dmatch)) ||
        (rc = io.out(summaries, P * 8, &dsum)) || (rc = io.out(lists, (size_t)list_cap * 4, &dlists)) || (rc = io.out(totals, 8, &dtot)))
        return rc;
    uint32_t *list_off = (uint32_t *)ctx->ws2;
    hipLaunchKernelGGL(pack_scan_kernel, dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, n_pairs, (const uint32_t *)dpid, (const int32_t *)dcnt,
                       threshold, (uint2 *)dsum, list_off, (uint32_t *)dtot);
    if (n_pairs > 0 && list_cap > 0)
        hipLaunchKernelGGL(pack_lists_kernel, dim3((n_pairs + 3) / 4), dim3(256), 0, ctx->stream, n_pairs, (const int32_t *)dcnt,
                           (const int32_t *)dmatch, stride, (const uint32_t *)list_off, (uint32_t *)dlists, list_cap,
                           (const FrameDev *)nullptr, (const int32_t *)nullptr);
    HIPCHK(ctx, hipGetLastError());
    return io.finish();
}

int orbm_allpairs_batch(orbm_ctx *ctx, int kind, int n_pairs, const int32_t *frame_a, const int32_t *frame_b,
                        const uint32_t *pair_id, float ratio, int th, int check_ori, int32_t stride, int threshold,
                        orbm_pair_summary *summaries, uint32_t *lists, uint32_t list_cap, uint32_t *totals, int io_on_device) {
    if (!ctx) return ORBM_EINVAL;
    if (kind < 0 || kind > 2) return orbm_fail(ctx, ORBM_EINVAL, "allpairs_batch: kind must be 0 (brute-force 2-NN), 1 (SearchByBoW KF,KF) or 2 (SearchByBoW KF,F)");
    if (n_pairs < 0 || stride < 0 || (n_pairs > 0 && (!frame_a || !frame_b || !summaries)) || !totals || (list_cap > 0 && !lists))
        return orbm_fail(ctx, ORBM_EINVAL, "allpairs_batch: null argument or negative size");
    if (stride > 65536) return orbm_fail(ctx, ORBM_ELIMIT, "allpairs_batch: stride %d: row and value indices must fit 16 bits", stride);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t P = (size_t)n_pairs, rows = P * (size_t)stride;
    int rc;
    if (!io_on_device) {
        // host lists can be checked (and the matcher's own staging must not run after ours: it shares the I/O workspace)
        rc = orbm_check_frames(ctx, n_pairs, frame_a, "frame_a");
        if (!rc) rc = orbm_check_frames(ctx, n_pairs, frame_b, "frame_b");
        if (rc) return rc;
        for (int p = 0; p < n_pairs; p++) {
            const int need = kind == 2 ? ctx->h_frames[frame_b[p]].n : ctx->h_frames[frame_a[p]].n;
            if (need > stride) return orbm_fail(ctx, ORBM_EINVAL, "allpairs_batch: stride %d < %d rows needed by pair %d", stride, need, p);
            // a list entry is (row | value << 16): the value indexes the OTHER frame of the pair, which the stride does not bound
            const int other = kind == 2 ? ctx->h_frames[frame_a[p]].n : ctx->h_frames[frame_b[p]].n;
            if (other > 65536)
                return orbm_fail(ctx, ORBM_ELIMIT, "allpairs_batch: pair %d matches into a frame of %d keypoints: value indices must fit 16 bits", p, other);
        }
    }
    if (kind == 0 && (rc = orbm_bf2nn_prepare(ctx))) return rc;
    if (kind != 0 && (rc = orbm_ensure_image(ctx, 5, 0, nullptr))) return rc;   // the BoW searches' operand image: built before anything is staged
    // match rows + counts stay in a workspace of their own (ws2 belongs to the matchers and to the packer)
    const size_t o_counts = (rows * 4 + 255) / 256 * 256, need3 = o_counts + P * 4 + 256;
    if (need3 > ctx->ws3_bytes) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->ws3) HIPCHK(ctx, hipFree(ctx->ws3));
        ctx->ws3 = nullptr; ctx->ws3_bytes = 0;
        HIPCHK(ctx, hipMalloc(&ctx->ws3, need3 + need3 / 8));
        ctx->ws3_bytes = need3 + need3 / 8;
    }
    int32_t *d_match = (int32_t *)ctx->ws3, *d_counts = (int32_t *)((char *)ctx->ws3 + o_counts);
    if ((rc = orbm_ws2_reserve(ctx, std::max(P * 4 + 256, P * 32 * sizeof(int))))) return rc;   // packer offsets / BoW split accumulators
    IoStage io(ctx, io_on_device);
    if (!io_on_device && (rc = orbm_ws_reserve(ctx, P * 20 + (size_t)list_cap * 4 + 4096))) return rc;
    const void *dfa, *dfb, *dpid; void *dsum, *dlists, *dtot;
    if ((rc = io.in(frame_a, P * 4, &dfa)) || (rc = io.in(frame_b, P * 4, &dfb)) || (rc = io.in(pair_id, P * 4, &dpid)) ||
        (rc = io.out(summaries, P * 8, &dsum)) || (rc = io.out(lists, (size_t)list_cap * 4, &dlists)) || (rc = io.out(totals, 8, &dtot)))
        return rc;
    if (n_pairs > 0) {
        if (kind == 0)
            rc = orbm_bf2nn_batch(ctx, n_pairs, (const int32_t *)dfa, (const int32_t *)dfb, ratio, th, stride, d_match, nullptr, nullptr, d_counts, 1);
        else
            rc = orbm_search_bow_batch(ctx, kind == 1 ? 1 : 0, n_pairs, (const int32_t *)dfa, (const int32_t *)dfb, ratio, check_ori, stride,
                                       d_match, d_counts, 1);
        if (rc) return rc;
    }
    // the BoW split path is done with its accumulators (stream order), the packer's offsets may reuse ws2
    uint32_t *list_off = (uint32_t *)ctx->ws2;
    hipLaunchKernelGGL(pack_scan_kernel, dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, n_pairs, (const uint32_t *)dpid, (const int32_t *)d_counts,
                       threshold, (uint2 *)dsum, list_off, (uint32_t *)dtot);
    if (n_pairs > 0 && list_cap > 0)
        hipLaunchKernelGGL(pack_lists_kernel, dim3((n_pairs + 3) / 4), dim3(256), 0, ctx->stream, n_pairs, (const int32_t *)d_counts,
                           (const int32_t *)d_match, stride, (const uint32_t *)list_off, (uint32_t *)dlists, list_cap,
                           (const FrameDev *)ctx->d_frames, (const int32_t *)(kind == 2 ? dfb : dfa));   // rows beyond the frame's n are never read
    HIPCHK(ctx, hipGetLastError());
    return io.finish();
}


// SURVEY 8(b) `orbm_allpairs(ctx[], n_gpus, ...)`: the pair loop of LoopClosing.cc:433-477 / Tracking.cc:1778-1801 over
// several GPUs from ONE host process.  Pairs are independent, so the list is cut into n_ctx contiguous slices, one worker
// thread per context runs orbm_allpairs_batch on its slice (host-memory I/O), and the records are concatenated in pair
// order: exactly the records one context would have produced for the whole list.  No collective: every context holds the
// keyframes its slice touches (handles[c][k] = handle of keyframe k in context c).
int orbm_allpairs(orbm_ctx *const *ctxs, int n_ctx, const int32_t *const *handles, int kind, int n_pairs, const int32_t *kf_a,
                  const int32_t *kf_b, const uint32_t *pair_id, float ratio, int th, int check_ori, int32_t stride, int threshold,
                  orbm_pair_summary *summaries, uint32_t *lists, uint32_t list_cap, uint32_t *totals) {
    if (!ctxs || n_ctx <= 0 || !ctxs[0]) return ORBM_EINVAL;
    orbm_ctx *c0 = ctxs[0];
    for (int c = 0; c < n_ctx; c++)
        if (!ctxs[c] || !handles || !handles[c]) return orbm_fail(c0, ORBM_EINVAL, "allpairs: context or handle table %d is null", c);
    if (n_pairs < 0 || stride < 0 || (n_pairs > 0 && (!kf_a || !kf_b || !summaries)) || !totals || (list_cap > 0 && !lists))
        return orbm_fail(c0, ORBM_EINVAL, "allpairs: null argument or negative size");
    totals[0] = totals[1] = 0;
    if (n_pairs == 0) return ORBM_OK;
    const int W = std::min(n_ctx, n_pairs);
    std::vector<int> first(W + 1);
    for (int c = 0; c <= W; c++) first[c] = (int)((long long)n_pairs * c / W);
    std::vector<int> rcs(W, ORBM_OK);
    std::vector<std::vector<uint32_t>> part_lists(W);
    std::vector<uint32_t> part_tot(2 * (size_t)W, 0);
    auto work = [&](int c) {
        const int p0 = first[c], np = first[c + 1] - p0;
        std::vector<int32_t> fa(np), fb(np);
        for (int p = 0; p < np; p++) { fa[p] = handles[c][kf_a[p0 + p]]; fb[p] = handles[c][kf_b[p0 + p]]; }
        std::vector<uint32_t> pid(np);
        for (int p = 0; p < np; p++) pid[p] = pair_id ? pair_id[p0 + p] : (uint32_t)(p0 + p);     // ids of the WHOLE list
        part_lists[c].resize(list_cap);
        rcs[c] = orbm_allpairs_batch(ctxs[c], kind, np, fa.data(), fb.data(), pid.data(), ratio, th, check_ori, stride, threshold,
                                     summaries + p0, list_cap ? part_lists[c].data() : nullptr, list_cap, &part_tot[2 * (size_t)c], 0);
    };
    std::vector<std::thread> workers;
    for (int c = 1; c < W; c++) workers.emplace_back(work, c);
    work(0);
    for (auto &t : workers) t.join();
    for (int c = 0; c < W; c++)
        if (rcs[c] < 0) {
            if (c) orbm_fail(c0, rcs[c], "allpairs: context %d: %s", c, orbm_last_error(ctxs[c]));
            return rcs[c];
        }
    uint64_t entries = 0;
    for (int c = 0; c < W; c++) {
        const uint32_t n = part_tot[2 * (size_t)c + 1];
        const uint32_t have = std::min(n, list_cap);                       // what the slice's buffer holds
        if (entries < list_cap && have) {
            const uint32_t take = (uint32_t)std::min<uint64_t>(have, list_cap - entries);
            memcpy(lists + entries, part_lists[c].data(), (size_t)take * 4);
        }
        entries += n;
        totals[0] += part_tot[2 * (size_t)c];
    }
    totals[1] = (uint32_t)std::min<uint64_t>(entries, 0xFFFFFFFFull);     // > list_cap: the lists were cut there, as orbm_allpairs_batch reports it
    return ORBM_OK;
}

}  // extern "C"
