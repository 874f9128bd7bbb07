// ctx.hip -- context, device-resident frame / query-set storage, grid-cell index build (a15, a16),
// batched DescriptorDistance (a1) and the popcount-rate micro-kernel (K0).
#include <algorithm>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "orbm_internal.h"

static thread_local std::string g_create_err;

int orbm_fail(orbm_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_err = buf;
    return code;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
constexpr size_t PIN_MAX_BYTES = (size_t)256 << 20;

int orbm_ws_reserve(orbm_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->ws_bytes) return ORBM_OK;
    if (ctx->ws) HIPCHK(ctx, hipFree(ctx->ws));
    ctx->ws = nullptr; ctx->ws_bytes = 0;
    size_t cap = align_up(bytes + bytes / 4, 1 << 20);
    HIPCHK(ctx, hipMalloc(&ctx->ws, cap));
    ctx->ws_bytes = cap;
    // pinned mirror (skipped for very large staging areas: those calls are bandwidth-, not latency-bound)
    if (ctx->pin) { (void)hipHostFree(ctx->pin); ctx->pin = nullptr; ctx->pin_bytes = 0; }
    if (cap <= PIN_MAX_BYTES && hipHostMalloc(&ctx->pin, cap, hipHostMallocDefault) == hipSuccess) ctx->pin_bytes = cap;
    else { ctx->pin = nullptr; (void)hipGetLastError(); }
    return ORBM_OK;
}

void *orbm_pin_up(orbm_ctx *ctx, size_t bytes) {
    if (bytes > PIN_MAX_BYTES) return nullptr;
    if (bytes <= ctx->pin_up_bytes) return ctx->pin_up;
    if (ctx->pin_up) { (void)hipHostFree(ctx->pin_up); ctx->pin_up = nullptr; ctx->pin_up_bytes = 0; }
    size_t cap = align_up(bytes + bytes / 4, 1 << 20);
    if (hipHostMalloc(&ctx->pin_up, cap, hipHostMallocDefault) != hipSuccess) { ctx->pin_up = nullptr; (void)hipGetLastError(); return nullptr; }
    ctx->pin_up_bytes = cap;
    return ctx->pin_up;
}

constexpr size_t ARENA_POOL_MAX_ENTRIES = 16;
constexpr size_t ARENA_POOL_MAX_BYTES = (size_t)256 << 20;

int orbm_arena_alloc(orbm_ctx *ctx, size_t bytes, Arena *out) {
    if (bytes == 0) bytes = 256;
    int best = -1;
    for (size_t i = 0; i < ctx->arena_pool.size(); i++) {
        const Arena &c = ctx->arena_pool[i];
        if (c.cap >= bytes && c.cap <= 4 * bytes + (1 << 20) && (best < 0 || c.cap < ctx->arena_pool[best].cap)) best = (int)i;
    }
    if (best >= 0) {
        *out = ctx->arena_pool[best];
        ctx->arena_pool_bytes -= out->cap;
        ctx->arena_pool.erase(ctx->arena_pool.begin() + best);
    } else {
        *out = Arena();
        out->cap = align_up(bytes, 4096);
        HIPCHK(ctx, hipMalloc(&out->base, out->cap));
    }
    out->bytes = bytes;
    out->live = 0;
    return ORBM_OK;
}

void orbm_arena_release(orbm_ctx *ctx, Arena &a) {
    if (!a.base) return;
    if (ctx->arena_pool.size() < ARENA_POOL_MAX_ENTRIES && ctx->arena_pool_bytes + a.cap <= ARENA_POOL_MAX_BYTES) {
        ctx->arena_pool.push_back(a);
        ctx->arena_pool_bytes += a.cap;
    } else {
        (void)hipFree(a.base);
    }
    a = Arena();
}

int orbm_ws2_reserve(orbm_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->ws2_bytes) return ORBM_OK;
    if (ctx->ws2) HIPCHK(ctx, hipFree(ctx->ws2));
    ctx->ws2 = nullptr; ctx->ws2_bytes = 0;
    size_t cap = align_up(bytes + bytes / 4, 1 << 20);
    HIPCHK(ctx, hipMalloc(&ctx->ws2, cap));
    ctx->ws2_bytes = cap;
    return ORBM_OK;
}

void *IoStage::carve(size_t bytes) {
    void *p = (char *)ctx->ws + used;
    used += align_up(bytes, 256);
    return p;
}

int IoStage::in(const void *p, size_t bytes, const void **devp) {
    if (dev || p == nullptr) { *devp = p; return ORBM_OK; }
    const size_t off = used;
    void *d = carve(bytes);
    if (used > ctx->ws_bytes) return orbm_fail(ctx, ORBM_ENOMEM, "I/O staging workspace overflow");
    if (ctx->pin && !in_sent) {           // staged: goes out with send_inputs() as one copy
        memcpy((char *)ctx->pin + off, p, bytes);
        in_end = used;
    } else {
        HIPCHK(ctx, hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    *devp = d;
    return ORBM_OK;
}

int IoStage::send_inputs() {
    if (dev || in_sent) return ORBM_OK;
    in_sent = true;
    if (in_end) HIPCHK(ctx, hipMemcpyAsync(ctx->ws, ctx->pin, in_end, hipMemcpyHostToDevice, ctx->stream));
    return ORBM_OK;
}

// every entry point registers its inputs first, then its outputs, then launches: the first out() sends the inputs
int IoStage::out(void *p, size_t bytes, void **devp) {
    if (!dev) { int rc = send_inputs(); if (rc) return rc; }
    if (dev || p == nullptr) { *devp = p; return ORBM_OK; }
    if (outs.empty()) out_begin = used;
    void *d = carve(bytes);
    if (used > ctx->ws_bytes) return orbm_fail(ctx, ORBM_ENOMEM, "I/O staging workspace overflow");
    outs.push_back({p, d, bytes});
    *devp = d;
    return ORBM_OK;
}

int IoStage::finish() {
    if (dev) return ORBM_OK;
    { int rc = send_inputs(); if (rc) return rc; }
    if (ctx->pin && !outs.empty()) {      // one D2H copy of the whole output region into the pinned mirror
        HIPCHK(ctx, hipMemcpyAsync((char *)ctx->pin + out_begin, (char *)ctx->ws + out_begin, used - out_begin,
                                   hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        for (auto &o : outs) memcpy(o.host, (char *)ctx->pin + ((char *)o.devp - (char *)ctx->ws), o.bytes);
        return ORBM_OK;
    }
    for (auto &o : outs) HIPCHK(ctx, hipMemcpyAsync(o.host, o.devp, o.bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return ORBM_OK;
}

int orbm_check_frames(orbm_ctx *ctx, int n, const int32_t *ids, const char *what) {
    for (int i = 0; i < n; i++) {
        int32_t id = ids[i];
        if (id < 0 || (size_t)id >= ctx->h_frames.size() || ctx->frame_arena[id] < 0)
            return orbm_fail(ctx, ORBM_EINVAL, "%s[%d] = %d is not a live frame id", what, i, id);
    }
    return ORBM_OK;
}

int orbm_check_qsets(orbm_ctx *ctx, int n, const int32_t *ids, int kind, const char *what) {
    for (int i = 0; i < n; i++) {
        int32_t id = ids[i];
        if (id < 0 || (size_t)id >= ctx->h_qsets.size() || ctx->qset_arena[id] < 0 || ctx->h_qsets[id].kind != kind)
            return orbm_fail(ctx, ORBM_EINVAL, "%s[%d] = %d is not a live query-set id of the right kind", what, i, id);
    }
    return ORBM_OK;
}

// ---------------------------------------------------------------------------------------------
// Grid-cell index build: Frame::AssignFeaturesToGrid + Frame::PosInGrid (Frame.cc:278-298, 506-519).
// One workgroup per frame: LDS histogram over the 3072 cells, block scan, scatter, then every cell's
// short list is put in ascending keypoint order (the reference appends in index order).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int pos_in_grid(const FrameG &f, int i) {
    const int posX = (int)roundf(__fmul_rn(__fsub_rn(f.x[i], f.min_x), f.inv_w));
    const int posY = (int)roundf(__fmul_rn(__fsub_rn(f.y[i], f.min_y), f.inv_h));
    if (posX < 0 || posX >= ORBM_GRID_COLS || posY < 0 || posY >= ORBM_GRID_ROWS) return -1;
    return posX * ORBM_GRID_ROWS + posY;
}

constexpr int GRID_LDS_MAX_N = 8192;   // keypoint lists up to this size are ordered in LDS

constexpr int GRID_THREADS = 1024;   // one workgroup per frame: the upload of every per-call search waits for it

__global__ __launch_bounds__(GRID_THREADS) void build_grid_kernel(const FrameDev *__restrict__ table,
                                                         const int32_t *__restrict__ ids) {
    __shared__ __align__(16) uint32_t cnt[ORBM_GRID_CELLS];
    __shared__ uint32_t off[ORBM_GRID_CELLS + 1];
    __shared__ uint32_t part[GRID_THREADS / 64];
    __shared__ uint16_t lidx[GRID_LDS_MAX_N];
    const FrameG f(table[ids[blockIdx.x]]);
    const int tid = threadIdx.x;
    const int n = f.n;
    const bool in_lds = n <= GRID_LDS_MAX_N;
    for (int c = tid; c < ORBM_GRID_CELLS; c += GRID_THREADS) cnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += GRID_THREADS) {
        const int c = pos_in_grid(f, i);
        if (c >= 0) atomicAdd(&cnt[c], 1u);
    }
    __syncthreads();
    // exclusive scan: PER consecutive cells per thread, a scan inside each wave, then the waves before this one
    constexpr int PER = ORBM_GRID_CELLS / GRID_THREADS;
    static_assert(PER * GRID_THREADS == ORBM_GRID_CELLS, "cells must divide evenly");
    uint32_t loc[PER];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) { loc[k] = s; s += cnt[tid * PER + k]; }
    uint32_t incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if ((tid & 63) >= d) incl += up;
    }
    if ((tid & 63) == 63) part[tid >> 6] = incl;
    __syncthreads();
    uint32_t base = incl - s;
    for (int w = 0; w < (tid >> 6); w++) base += part[w];
    if (tid == GRID_THREADS - 1) { off[ORBM_GRID_CELLS] = base + s; f.cell_off[ORBM_GRID_CELLS] = base + s; }
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const uint32_t o = base + loc[k];
        off[tid * PER + k] = o;
        f.cell_off[tid * PER + k] = o;
        cnt[tid * PER + k] = o;  // becomes the scatter cursor
    }
    __syncthreads();
    for (int i = tid; i < n; i += GRID_THREADS) {
        const int c = pos_in_grid(f, i);
        if (c < 0) continue;
        const uint32_t at = atomicAdd(&cnt[c], 1u);
        if (in_lds) lidx[at] = (uint16_t)i; else f.cell_idx[at] = (uint32_t)i;
    }
    __syncthreads();
    if (!in_lds) __threadfence_block();
    // ascending order inside each cell (lists are short: ~0.65 keypoints per cell at n=2000)
    for (int c = tid; c < ORBM_GRID_CELLS; c += GRID_THREADS) {
        const uint32_t b = off[c], e = cnt[c];
        if (in_lds) {
            for (uint32_t i = b + 1; i < e; i++) {
                const uint16_t v = lidx[i];
                uint32_t j = i;
                while (j > b && lidx[j - 1] > v) { lidx[j] = lidx[j - 1]; j--; }
                lidx[j] = v;
            }
        } else {
            for (uint32_t i = b + 1; i < e; i++) {
                const uint32_t v = f.cell_idx[i];
                uint32_t j = i;
                while (j > b && f.cell_idx[j - 1] > v) { f.cell_idx[j] = f.cell_idx[j - 1]; j--; }
                f.cell_idx[j] = v;
            }
        }
    }
    __syncthreads();
    if (!in_lds) __threadfence_block();
    // cell-ordered records, one per list position (independent loads)
    const uint32_t total = off[ORBM_GRID_CELLS];
    for (uint32_t i = tid; i < total; i += GRID_THREADS) {
        const uint32_t k = in_lds ? (uint32_t)lidx[i] : f.cell_idx[i];
        if (in_lds) f.cell_idx[i] = k;
        const uint32_t c = (uint32_t)pos_in_grid(f, (int)k);                      // ix * 48 + iy: the cell the record sits in
        const uint32_t cx = c / ORBM_GRID_ROWS, cy = c - cx * ORBM_GRID_ROWS;
        ((gptr<v4f>)f.cell_rec)[i] = v4f{f.x[k], f.y[k], f.uright[k],
                                         __uint_as_float(k | ((uint32_t)f.octave[k] << 16) | (cy << 20) | (cx << 26))};
    }
}

// a1: n independent 256-bit distances; one lane per pair, rows read as two 16-byte loads.
__global__ __launch_bounds__(256) void hamming_pairs_kernel(int n, const uint4 *__restrict__ a,
                                                            const uint4 *__restrict__ b, int32_t *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint4 a0 = a[2 * i], a1 = a[2 * i + 1], b0 = b[2 * i], b1 = b[2 * i + 1];
    const uint32_t x[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const uint32_t y[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    out[i] = (int32_t)hamming256(x, y);
}

// K0: sustained xor + bcnt(accumulate) issue rate.  Every lane runs `iters` rounds of 8 independent
// xor/bcnt pairs (the shape of one DescriptorDistance); the result is stored so nothing is dead.
__global__ __launch_bounds__(256) void popcount_peak_kernel(int iters, uint32_t seed, uint32_t *__restrict__ sink) {
    uint32_t q[8], acc[4] = {0, 0, 0, 0};
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; i++) q[i] = (t + 1) * 2654435761u + i * 40503u + seed;
    acc[0] = seed; acc[1] = seed * 3u; acc[2] = seed * 5u; acc[3] = seed * 7u;
    for (int it = 0; it < iters; it++) {
        // 8 steps x 4 independent (v_xor_b32 + v_bcnt_u32_b32 accumulate) pairs, all operands in VGPRs;
        // each pair reads a neighbour accumulator so that nothing is loop-invariant.
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t b0 = acc[1], b1 = acc[2], b2 = acc[3], b3 = acc[0];
            acc[0] = __builtin_popcount(q[i] ^ b0) + acc[0];
            acc[1] = __builtin_popcount(q[i] ^ b1) + acc[1];
            acc[2] = __builtin_popcount(q[i] ^ b2) + acc[2];
            acc[3] = __builtin_popcount(q[i] ^ b3) + acc[3];
        }
    }
    sink[t] = acc[0] + acc[1] + acc[2] + acc[3];
}

hipEvent_t *orbm_timing_slot(orbm_ctx *ctx) {
    if ((size_t)(2 * ctx->tev_used + 2) > ctx->tev.size()) return nullptr;
    return &ctx->tev[2 * (size_t)ctx->tev_used++];
}

extern "C" {

orbm_ctx *orbm_create(int device) {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        orbm_fail(nullptr, ORBM_EHIP, "no HIP device available (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return nullptr;
    }
    if (device < 0 || device >= ndev) {
        orbm_fail(nullptr, ORBM_EINVAL, "device %d out of range (have %d)", device, ndev);
        return nullptr;
    }
    orbm_ctx *ctx = new (std::nothrow) orbm_ctx();
    if (!ctx) { orbm_fail(nullptr, ORBM_ENOMEM, "out of host memory"); return nullptr; }
    ctx->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipGetDeviceProperties(&ctx->prop, device)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) {
        orbm_fail(nullptr, ORBM_EHIP, "context set-up failed: %s", hipGetErrorString(e));
        delete ctx;
        return nullptr;
    }
    ctx->stream = ctx->own_stream;
    ctx->err.reserve(512);                     // orbm_fail writes at most 512 bytes: no host allocation on an error path later
    if (orbm_bow_scratch_resolve(ctx) != ORBM_OK) {
        g_create_err = ctx->err;
        (void)hipStreamDestroy(ctx->own_stream);
        delete ctx;
        return nullptr;
    }
    return ctx;
}

int orbm_kernel_timing(orbm_ctx *ctx, int max_launches) {
    if (!ctx) return ORBM_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (hipEvent_t e : ctx->tev) (void)hipEventDestroy(e);
    ctx->tev.clear(); ctx->tev_used = 0;
    if (max_launches <= 0) return ORBM_OK;
    if (max_launches > 65536) return orbm_fail(ctx, ORBM_ELIMIT, "kernel_timing: at most 65536 launches");
    for (int i = 0; i < 2 * max_launches; i++) {
        hipEvent_t e;
        HIPCHK(ctx, hipEventCreate(&e));
        ctx->tev.push_back(e);
    }
    return ORBM_OK;
}

int orbm_kernel_times(orbm_ctx *ctx, float *ms, int cap) {
    if (!ctx || (cap > 0 && !ms)) return ORBM_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const int n = std::min(ctx->tev_used, cap);
    for (int i = 0; i < n; i++) HIPCHK(ctx, hipEventElapsedTime(&ms[i], ctx->tev[2 * i], ctx->tev[2 * i + 1]));
    const int used = ctx->tev_used;
    ctx->tev_used = 0;
    return used;
}

void orbm_destroy(orbm_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (hipEvent_t e : ctx->tev) (void)hipEventDestroy(e);
    for (auto &a : ctx->arenas) if (a.base) (void)hipFree(a.base);
    for (auto &a : ctx->arena_pool) if (a.base) (void)hipFree(a.base);
    for (auto &v : ctx->vocabs) if (v.arena.base) (void)hipFree(v.arena.base);
    for (auto &a : ctx->bvec_arenas) if (a.base) (void)hipFree(a.base);
    if (ctx->d_bvecs) (void)hipFree(ctx->d_bvecs);
    if (ctx->d_frames) (void)hipFree(ctx->d_frames);
    if (ctx->d_qsets) (void)hipFree(ctx->d_qsets);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    if (ctx->pin_up) (void)hipHostFree(ctx->pin_up);
    if (ctx->ws2) (void)hipFree(ctx->ws2);
    if (ctx->ws3) (void)hipFree(ctx->ws3);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char *orbm_last_error(const orbm_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int orbm_set_stream(orbm_ctx *ctx, void *hip_stream) {
    if (!ctx) return ORBM_EINVAL;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream == (void *)-1 ? ctx->own_stream : (hipStream_t)hip_stream;
    return ORBM_OK;
}

int orbm_synchronize(orbm_ctx *ctx) {
    if (!ctx) return ORBM_EINVAL;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return ORBM_OK;
}

int orbm_device_info(orbm_ctx *ctx, int32_t *n_cu, int32_t *clock_khz, char *name, int name_len) {
    if (!ctx) return ORBM_EINVAL;
    if (n_cu) *n_cu = ctx->prop.multiProcessorCount;
    if (clock_khz) *clock_khz = ctx->prop.clockRate;
    if (name && name_len > 0) snprintf(name, name_len, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
    return ORBM_OK;
}

static int grow_table(orbm_ctx *ctx, void **d_tab, size_t *cap, size_t need, size_t elem) {
    if (need <= *cap) return ORBM_OK;
    size_t ncap = *cap ? *cap : 1024;
    while (ncap < need) ncap *= 2;
    void *nt = nullptr;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMalloc(&nt, ncap * elem));
    if (*d_tab) HIPCHK(ctx, hipFree(*d_tab));
    *d_tab = nt;
    *cap = ncap;
    return ORBM_OK;
}

struct Packer {
    std::vector<uint8_t> host;
    size_t off = 0;
    size_t add(const void *src, size_t bytes, size_t cap = 0) {   // src may be null: reserved, zero-filled; cap >= bytes reserves room
        size_t o = off;
        off = align_up(off + std::max(bytes, cap), 256);
        host.resize(off, 0);
        if (src && bytes) memcpy(host.data() + o, src, bytes);
        return o;
    }
};

int orbm_frames_upload(orbm_ctx *ctx, int n_frames, const orbm_frame_view *views, int32_t *out_ids) {
    if (!ctx) return ORBM_EINVAL;
    if (n_frames < 0 || (n_frames > 0 && (!views || !out_ids))) return orbm_fail(ctx, ORBM_EINVAL, "frames_upload: null argument");
    if (n_frames == 0) return ORBM_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    struct Offs { size_t desc, x, y, angle, angle_raw, uright, octave, flags, cell_off, cell_idx, cell_rec, cell_nxt, fv_node, fv_off, fv_idx; bool has_nxt; };
    std::vector<Offs> offs(n_frames);
    Packer pk;
    for (int k = 0; k < n_frames; k++) {
        const orbm_frame_view &v = views[k];
        if (v.n < 0 || v.n > 65535) return orbm_fail(ctx, ORBM_ELIMIT, "frame %d: n = %d outside [0, 65535]", k, v.n);
        if (v.n > 0 && (!v.desc || !v.x || !v.y || !v.angle || !v.octave))
            return orbm_fail(ctx, ORBM_EINVAL, "frame %d: desc/x/y/angle/octave must be non-null", k);
        if (v.n_levels < 1 || v.n_levels > ORBM_MAX_LEVELS || !v.scale_factors || !v.level_sigma2)
            return orbm_fail(ctx, ORBM_EINVAL, "frame %d: n_levels = %d outside [1, %d] or null scale tables", k, v.n_levels, ORBM_MAX_LEVELS);
        if (v.n_nodes < 0 || (v.n_nodes > 0 && (!v.fv_node || !v.fv_off || !v.fv_idx)))
            return orbm_fail(ctx, ORBM_EINVAL, "frame %d: bad FeatureVector", k);
        for (int i = 0; i < v.n; i++)
            if (v.octave[i] < 0 || v.octave[i] >= v.n_levels)
                return orbm_fail(ctx, ORBM_EINVAL, "frame %d: octave[%d] = %d outside [0, %d)", k, i, v.octave[i], v.n_levels);
        uint32_t nfv = v.n_nodes > 0 ? v.fv_off[v.n_nodes] : 0;
        for (int j = 0; j < v.n_nodes; j++) {
            if (j > 0 && v.fv_node[j] <= v.fv_node[j - 1]) return orbm_fail(ctx, ORBM_EINVAL, "frame %d: fv_node not strictly ascending", k);
            if (v.fv_off[j + 1] < v.fv_off[j]) return orbm_fail(ctx, ORBM_EINVAL, "frame %d: fv_off not monotone", k);
        }
        for (uint32_t j = 0; j < nfv; j++)
            if (v.fv_idx[j] >= (uint32_t)v.n) return orbm_fail(ctx, ORBM_EINVAL, "frame %d: fv_idx[%u] out of range", k, j);
        Offs &o = offs[k];
        const size_t n = (size_t)v.n;
        o.desc = pk.add(v.desc, n * 32);
        o.x = pk.add(v.x, n * 4);
        o.y = pk.add(v.y, n * 4);
        o.angle = pk.add(v.angle, n * 4);
        o.angle_raw = v.angle_raw ? pk.add(v.angle_raw, n * 4) : o.angle;
        o.uright = pk.add(v.uright, n * 4);
        if (!v.uright) { float *u = (float *)(pk.host.data() + o.uright); for (size_t i = 0; i < n; i++) u[i] = -1.0f; }
        o.octave = pk.add(nullptr, n);
        for (size_t i = 0; i < n; i++) pk.host[o.octave + i] = (uint8_t)v.octave[i];
        o.flags = pk.add(v.mp_flags, n);
        o.cell_off = pk.add(nullptr, (ORBM_GRID_CELLS + 1) * 4);
        o.cell_idx = pk.add(nullptr, (n ? n : 1) * 4);
        o.cell_rec = pk.add(nullptr, (n ? n : 1) * 16);
        o.has_nxt = false;          // the skip lists of round 1 are gone: collect_kernel builds its own per-launch index
        o.cell_nxt = 0;
        // room for a FeatureVector computed on the device later (orbm_frames_compute_bow): at most one node per feature
        o.fv_node = pk.add(v.fv_node, (size_t)v.n_nodes * 4, n * 4);
        o.fv_off = pk.add(v.fv_off, ((size_t)v.n_nodes + 1) * 4, (n + 1) * 4);
        o.fv_idx = pk.add(v.fv_idx, (size_t)nfv * 4, n * 4);
    }
    Arena ar;
    { int rc = orbm_arena_alloc(ctx, pk.off, &ar); if (rc) return rc; }
    ar.live = n_frames;
    int arena_id = -1;
    for (size_t i = 0; i < ctx->arenas.size(); i++) if (!ctx->arenas[i].base) { arena_id = (int)i; break; }
    if (arena_id < 0) { ctx->arenas.push_back(ar); arena_id = (int)ctx->arenas.size() - 1; } else ctx->arenas[arena_id] = ar;

    char *b = (char *)ar.base;
    size_t lo = SIZE_MAX, hi = 0;
    for (int k = 0; k < n_frames; k++) {
        const orbm_frame_view &v = views[k];
        const Offs &o = offs[k];
        FrameDev f;
        memset(&f, 0, sizeof f);
        f.n = v.n; f.n_nodes = v.n_nodes; f.n_levels = v.n_levels;
        f.desc = (const uint32_t *)(b + o.desc);
        f.x = (const float *)(b + o.x); f.y = (const float *)(b + o.y);
        f.angle = (const float *)(b + o.angle); f.angle_raw = (const float *)(b + o.angle_raw);
        f.uright = (const float *)(b + o.uright);
        f.octave = (const uint8_t *)(b + o.octave);
        f.mp_flags = (uint8_t *)(b + o.flags);
        f.cell_off = (uint32_t *)(b + o.cell_off); f.cell_idx = (uint32_t *)(b + o.cell_idx);
        f.cell_rec = (float4 *)(b + o.cell_rec);
        f.cell_nxt = o.has_nxt ? (uint16_t *)(b + o.cell_nxt) : nullptr;
        f.fv_node = (const uint32_t *)(b + o.fv_node); f.fv_off = (const uint32_t *)(b + o.fv_off);
        f.fv_idx = (const uint32_t *)(b + o.fv_idx);
        f.min_x = v.min_x; f.min_y = v.min_y; f.max_x = v.max_x; f.max_y = v.max_y;
        f.inv_w = v.inv_w; f.inv_h = v.inv_h;
        for (int l = 0; l < ORBM_MAX_LEVELS; l++) {
            f.sf[l] = l < v.n_levels ? v.scale_factors[l] : 0.f;
            f.sigma2[l] = l < v.n_levels ? v.level_sigma2[l] : 0.f;
        }
        int32_t id;
        if (!ctx->free_frames.empty()) { id = ctx->free_frames.back(); ctx->free_frames.pop_back(); ctx->h_frames[id] = f; ctx->frame_arena[id] = arena_id; ctx->frame_arena8[id] = -1; ctx->frame_arena4[id] = -1; ctx->frame_arena4f[id] = -1; }
        else { id = (int32_t)ctx->h_frames.size(); ctx->h_frames.push_back(f); ctx->frame_arena.push_back(arena_id); ctx->frame_arena8.push_back(-1); ctx->frame_arena4.push_back(-1); ctx->frame_arena4f.push_back(-1); }
        out_ids[k] = id;
        lo = std::min(lo, (size_t)id); hi = std::max(hi, (size_t)id);
    }
    size_t old_cap = ctx->d_frames_cap;
    int rc = grow_table(ctx, (void **)&ctx->d_frames, &ctx->d_frames_cap, ctx->h_frames.size(), sizeof(FrameDev));
    if (rc) return rc;
    if (ctx->d_frames_cap != old_cap) { lo = 0; hi = ctx->h_frames.size() - 1; }
    rc = orbm_ws_reserve(ctx, (size_t)n_frames * 4);
    if (rc) return rc;
    // data, table slice and id list go through one pinned block when it fits (three stream-ordered copies, one sync)
    const size_t tab_bytes = (hi - lo + 1) * sizeof(FrameDev), id_bytes = (size_t)n_frames * 4;
    const size_t o_tab = align_up(pk.off, 256), o_ids = o_tab + align_up(tab_bytes, 256);
    if (char *pin = (char *)orbm_pin_up(ctx, o_ids + id_bytes)) {
        memcpy(pin, pk.host.data(), pk.off);
        memcpy(pin + o_tab, ctx->h_frames.data() + lo, tab_bytes);
        memcpy(pin + o_ids, out_ids, id_bytes);
        HIPCHK(ctx, hipMemcpyAsync(ar.base, pin, pk.off, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_frames + lo, pin + o_tab, tab_bytes, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(ctx->ws, pin + o_ids, id_bytes, hipMemcpyHostToDevice, ctx->stream));
    } else {
        HIPCHK(ctx, hipMemcpy(ar.base, pk.host.data(), pk.off, hipMemcpyHostToDevice));
        HIPCHK(ctx, hipMemcpy(ctx->d_frames + lo, ctx->h_frames.data() + lo, tab_bytes, hipMemcpyHostToDevice));
        HIPCHK(ctx, hipMemcpyAsync(ctx->ws, out_ids, id_bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    // build the grid-cell CSR on the device
    hipLaunchKernelGGL(build_grid_kernel, dim3(n_frames), dim3(GRID_THREADS), 0, ctx->stream, ctx->d_frames, (const int32_t *)ctx->ws);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return ORBM_OK;
}

static void arena_unref(orbm_ctx *ctx, int a) {
    if (a < 0) return;
    if (--ctx->arenas[a].live == 0) orbm_arena_release(ctx, ctx->arenas[a]);
}

int orbm_frames_free(orbm_ctx *ctx, int n_frames, const int32_t *ids) {
    if (!ctx) return ORBM_EINVAL;
    int rc = orbm_check_frames(ctx, n_frames, ids, "ids");
    if (rc) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n_frames; i++) {
        int32_t id = ids[i];
        if (ctx->frame_arena[id] < 0) continue;  // duplicate in the list
        arena_unref(ctx, ctx->frame_arena[id]);
        ctx->frame_arena[id] = -1;
        arena_unref(ctx, ctx->frame_arena8[id]);
        ctx->frame_arena8[id] = -1;
        arena_unref(ctx, ctx->frame_arena4[id]);
        ctx->frame_arena4[id] = -1;
        arena_unref(ctx, ctx->frame_arena4f[id]);
        ctx->frame_arena4f[id] = -1;
        ctx->free_frames.push_back(id);
    }
    return ORBM_OK;
}

int orbm_frame_set_mp_flags(orbm_ctx *ctx, int32_t id, const uint8_t *mp_flags) {
    if (!ctx) return ORBM_EINVAL;
    int rc = orbm_check_frames(ctx, 1, &id, "id");
    if (rc) return rc;
    if (!mp_flags) return orbm_fail(ctx, ORBM_EINVAL, "mp_flags is null");
    const FrameDev &f = ctx->h_frames[id];
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // through the pinned mirror when it is large enough (a copy from pageable memory is staged by the runtime, at twice the cost)
    const void *src = mp_flags;
    if (ctx->pin && (size_t)f.n <= ctx->pin_bytes) { memcpy(ctx->pin, mp_flags, (size_t)f.n); src = ctx->pin; }
    HIPCHK(ctx, hipMemcpyAsync(f.mp_flags, src, (size_t)f.n, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return ORBM_OK;
}

int orbm_frame_grid_download(orbm_ctx *ctx, int32_t id, uint32_t *cell_off, uint32_t *cell_idx) {
    if (!ctx) return ORBM_EINVAL;
    int rc = orbm_check_frames(ctx, 1, &id, "id");
    if (rc) return rc;
    const FrameDev &f = ctx->h_frames[id];
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (cell_off) HIPCHK(ctx, hipMemcpy(cell_off, f.cell_off, (ORBM_GRID_CELLS + 1) * 4, hipMemcpyDeviceToHost));
    if (cell_idx && f.n > 0) HIPCHK(ctx, hipMemcpy(cell_idx, f.cell_idx, (size_t)f.n * 4, hipMemcpyDeviceToHost));
    return ORBM_OK;
}

static int upload_qsets(orbm_ctx *ctx, int n_sets, const orbm_proj_queries *ps, const orbm_win_queries *wsets, int32_t *out_ids) {
    if (n_sets < 0 || (n_sets > 0 && ((!ps && !wsets) || !out_ids))) return orbm_fail(ctx, ORBM_EINVAL, "queries_upload: null argument");
    if (n_sets == 0) return ORBM_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    struct Offs { size_t desc, f0, f1, f2, f3, angle, l0, l1, flags; bool has_f3, has_angle; };
    std::vector<Offs> offs(n_sets);
    std::vector<uint32_t> packed;
    Packer pk;
    for (int k = 0; k < n_sets; k++) {
        Offs &o = offs[k];
        if (ps) {
            const orbm_proj_queries &q = ps[k];
            if (q.n < 0) return orbm_fail(ctx, ORBM_EINVAL, "query set %d: n < 0", k);
            if (q.n > 0 && (!q.desc || !q.proj_x || !q.proj_y || !q.proj_xr || !q.level || !q.view_cos))
                return orbm_fail(ctx, ORBM_EINVAL, "query set %d: null array", k);
            const size_t n = (size_t)q.n;
            for (size_t i = 0; i < n; i++)
                if (q.level[i] < 0 || q.level[i] >= ORBM_MAX_LEVELS) return orbm_fail(ctx, ORBM_EINVAL, "query set %d: level[%zu] out of range", k, i);
            o.desc = pk.add(q.desc, n * 32);
            // the six parameter arrays become one 16-byte record per query (QSetDev::pk)
            packed.resize(4 * n);
            for (size_t i = 0; i < n; i++) {
                memcpy(&packed[4 * i], &q.proj_x[i], 4); memcpy(&packed[4 * i + 1], &q.proj_y[i], 4); memcpy(&packed[4 * i + 2], &q.proj_xr[i], 4);
                const uint32_t fl = q.flags ? q.flags[i] : (uint32_t)(ORBM_Q_VALID | ORBM_Q_OBS);
                packed[4 * i + 3] = (uint32_t)q.level[i] | (fl << 8) | (q.view_cos[i] >= 0.998f ? 1u << 16 : 0u);   // ORBmatcher.cc:168-174
            }
            o.f0 = pk.add(packed.data(), n * 16);
            o.f1 = o.f2 = o.f3 = o.l0 = o.l1 = o.flags = 0; o.has_f3 = false; o.has_angle = false; o.angle = 0;
        } else {
            const orbm_win_queries &q = wsets[k];
            if (q.n < 0) return orbm_fail(ctx, ORBM_EINVAL, "query set %d: n < 0", k);
            if (q.n > 0 && (!q.desc || !q.u || !q.v || !q.radius || !q.min_level || !q.max_level))
                return orbm_fail(ctx, ORBM_EINVAL, "query set %d: null array", k);
            const size_t n = (size_t)q.n;
            o.desc = pk.add(q.desc, n * 32);
            o.f0 = pk.add(q.u, n * 4); o.f1 = pk.add(q.v, n * 4); o.f2 = pk.add(q.radius, n * 4);
            o.has_f3 = q.ur != nullptr; o.f3 = o.has_f3 ? pk.add(q.ur, n * 4) : 0;
            o.has_angle = q.angle != nullptr; o.angle = o.has_angle ? pk.add(q.angle, n * 4) : 0;
            o.l0 = pk.add(q.min_level, n * 4); o.l1 = pk.add(q.max_level, n * 4);
            o.flags = pk.add(q.flags, n);
            if (!q.flags) memset(pk.host.data() + o.flags, ORBM_Q_VALID | ORBM_Q_OBS, n);
        }
    }
    Arena ar;
    { int rc = orbm_arena_alloc(ctx, pk.off, &ar); if (rc) return rc; }
    ar.live = n_sets;
    int arena_id = -1;
    for (size_t i = 0; i < ctx->arenas.size(); i++) if (!ctx->arenas[i].base) { arena_id = (int)i; break; }
    if (arena_id < 0) { ctx->arenas.push_back(ar); arena_id = (int)ctx->arenas.size() - 1; } else ctx->arenas[arena_id] = ar;
    char *b = (char *)ar.base;
    size_t lo = SIZE_MAX, hi = 0;
    for (int k = 0; k < n_sets; k++) {
        const Offs &o = offs[k];
        QSetDev q;
        memset(&q, 0, sizeof q);
        q.n = ps ? ps[k].n : wsets[k].n;
        q.kind = ps ? 0 : 1;
        q.desc = (const uint32_t *)(b + o.desc);
        if (ps) {
            q.pk = (const uint32_t *)(b + o.f0);
        } else {
            q.f0 = (const float *)(b + o.f0); q.f1 = (const float *)(b + o.f1); q.f2 = (const float *)(b + o.f2);
            q.f3 = o.has_f3 ? (const float *)(b + o.f3) : nullptr;
            q.angle = o.has_angle ? (const float *)(b + o.angle) : nullptr;
            q.l0 = (const int32_t *)(b + o.l0); q.l1 = (const int32_t *)(b + o.l1);
            q.flags = (const uint8_t *)(b + o.flags);
        }
        int32_t id;
        if (!ctx->free_qsets.empty()) { id = ctx->free_qsets.back(); ctx->free_qsets.pop_back(); ctx->h_qsets[id] = q; ctx->qset_arena[id] = arena_id; }
        else { id = (int32_t)ctx->h_qsets.size(); ctx->h_qsets.push_back(q); ctx->qset_arena.push_back(arena_id); }
        out_ids[k] = id;
        lo = std::min(lo, (size_t)id); hi = std::max(hi, (size_t)id);
    }
    size_t old_cap = ctx->d_qsets_cap;
    int rc = grow_table(ctx, (void **)&ctx->d_qsets, &ctx->d_qsets_cap, ctx->h_qsets.size(), sizeof(QSetDev));
    if (rc) return rc;
    if (ctx->d_qsets_cap != old_cap) { lo = 0; hi = ctx->h_qsets.size() - 1; }
    const size_t tab_bytes = (hi - lo + 1) * sizeof(QSetDev), o_tab = align_up(pk.off, 256);
    if (char *pin = (char *)orbm_pin_up(ctx, o_tab + tab_bytes)) {
        // stream-ordered pinned copies: the searches that follow on this context's stream see the data
        if (pk.off) memcpy(pin, pk.host.data(), pk.off);
        memcpy(pin + o_tab, ctx->h_qsets.data() + lo, tab_bytes);
        if (pk.off) HIPCHK(ctx, hipMemcpyAsync(ar.base, pin, pk.off, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_qsets + lo, pin + o_tab, tab_bytes, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // the pinned block is reused by the next upload
    } else {
        if (pk.off) HIPCHK(ctx, hipMemcpy(ar.base, pk.host.data(), pk.off, hipMemcpyHostToDevice));
        HIPCHK(ctx, hipMemcpy(ctx->d_qsets + lo, ctx->h_qsets.data() + lo, tab_bytes, hipMemcpyHostToDevice));
    }
    return ORBM_OK;
}

int orbm_proj_queries_upload(orbm_ctx *ctx, int n_sets, const orbm_proj_queries *sets, int32_t *out_ids) {
    if (!ctx) return ORBM_EINVAL;
    return upload_qsets(ctx, n_sets, sets, nullptr, out_ids);
}

int orbm_win_queries_upload(orbm_ctx *ctx, int n_sets, const orbm_win_queries *sets, int32_t *out_ids) {
    if (!ctx) return ORBM_EINVAL;
    return upload_qsets(ctx, n_sets, nullptr, sets, out_ids);
}

int orbm_queries_free(orbm_ctx *ctx, int n_sets, const int32_t *ids) {
    if (!ctx) return ORBM_EINVAL;
    for (int i = 0; i < n_sets; i++) {
        int32_t id = ids[i];
        if (id < 0 || (size_t)id >= ctx->h_qsets.size()) return orbm_fail(ctx, ORBM_EINVAL, "ids[%d] = %d out of range", i, id);
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n_sets; i++) {
        int32_t id = ids[i];
        if (ctx->qset_arena[id] < 0) continue;
        arena_unref(ctx, ctx->qset_arena[id]);
        ctx->qset_arena[id] = -1;
        ctx->free_qsets.push_back(id);
    }
    return ORBM_OK;
}

int orbm_hamming256_pairs(orbm_ctx *ctx, int n, const uint8_t *a, const uint8_t *b, int32_t *dist, int io_on_device) {
    if (!ctx) return ORBM_EINVAL;
    if (n < 0 || (n > 0 && (!a || !b || !dist))) return orbm_fail(ctx, ORBM_EINVAL, "hamming256_pairs: null argument");
    if (n == 0) return ORBM_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (io_on_device && ((((uintptr_t)a) | ((uintptr_t)b)) & 15))
        return orbm_fail(ctx, ORBM_EINVAL, "hamming256_pairs: device rows must be 16-byte aligned");
    IoStage io(ctx, io_on_device);
    if (!io_on_device) { int rc = orbm_ws_reserve(ctx, (size_t)n * 68 + 1024); if (rc) return rc; }
    const void *da, *db; void *dd;
    int rc;
    if ((rc = io.in(a, (size_t)n * 32, &da)) || (rc = io.in(b, (size_t)n * 32, &db)) || (rc = io.out(dist, (size_t)n * 4, &dd))) return rc;
    hipLaunchKernelGGL(hamming_pairs_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, (const uint4 *)da, (const uint4 *)db, (int32_t *)dd);
    HIPCHK(ctx, hipGetLastError());
    return io.finish();
}

int orbm_popcount_peak(orbm_ctx *ctx, int iters, double *lane_ops_per_s) {
    if (!ctx || !lane_ops_per_s || iters <= 0) return ORBM_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int blocks = ctx->prop.multiProcessorCount * 8;  // 8 x 256 threads = 32 waves per CU
    int rc = orbm_ws2_reserve(ctx, (size_t)blocks * 256 * 4);
    if (rc) return rc;
    hipEvent_t e0, e1;
    HIPCHK(ctx, hipEventCreate(&e0));
    HIPCHK(ctx, hipEventCreate(&e1));
    hipLaunchKernelGGL(popcount_peak_kernel, dim3(blocks), dim3(256), 0, ctx->stream, iters / 8 + 1, 12345u, (uint32_t *)ctx->ws2);  // warm-up
    HIPCHK(ctx, hipEventRecord(e0, ctx->stream));
    hipLaunchKernelGGL(popcount_peak_kernel, dim3(blocks), dim3(256), 0, ctx->stream, iters, 777u, (uint32_t *)ctx->ws2);
    HIPCHK(ctx, hipEventRecord(e1, ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(ctx, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    // per iteration and lane: 4 x 8 x (xor + bcnt) = 64 lane-ops (the scalar LCG is not counted)
    *lane_ops_per_s = (double)blocks * 256.0 * (double)iters * 64.0 / ((double)ms * 1e-3);
    return ORBM_OK;
}

}  // extern "C"

// ---- +-1 images of the descriptors (operands of the MFMA brute-force search) -------------------------------------------
// BITS = 8: every descriptor bit as one int8 (0 -> +1, 1 -> -1), 256 B per row.  BITS = 4: as FP4 E2M1 (+1.0 = 0x2,
// -1.0 = 0xA), 128 B per row.  thread = one 16-byte chunk of one output row; rows >= n -> 0 (rows padded to 32).
template <int BITS>
__global__ __launch_bounds__(256) void expand_image_kernel(FrameDev *__restrict__ table, const int32_t *__restrict__ ids, int blocks_per_frame) {
    constexpr int CHUNKS = BITS == 8 ? 16 : 8;          // 16-byte chunks per output row
    const int f = blockIdx.x / blocks_per_frame;
    const FrameG F(table[ids[f]]);
    const int row = (blockIdx.x - f * blocks_per_frame) * (256 / CHUNKS) + (int)(threadIdx.x / CHUNKS), chunk = threadIdx.x % CHUNKS;
    const int n8 = max(32, (F.n + 31) & ~31);
    if (row >= n8) return;
    v4u out = {0, 0, 0, 0};
    if (row < F.n) {
        uint32_t o[4];
        if (BITS == 8) {
            const uint32_t w = F.desc[(size_t)row * 8 + (chunk >> 1)];
            const uint32_t bits = (chunk & 1) ? (w >> 16) : (w & 0xFFFFu);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t nib = (bits >> (4 * q)) & 0xFu;
                const uint32_t b01 = (nib * 0x00204081u) & 0x01010101u;   // bit k of the nibble -> byte k
                o[q] = 0x01010101u + b01 * 0xFEu;                          // byte 0 -> 0x01 (+1), byte 1 -> 0xFF (-1); no carries
            }
        } else {
            const uint32_t w = F.desc[(size_t)row * 8 + chunk];            // 32 bits -> 32 nibbles = 16 bytes
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t byte = (w >> (8 * q)) & 0xFFu;
                uint32_t v = 0x22222222u;                                  // +1.0 everywhere
#pragma unroll
                for (int j = 0; j < 8; j++) v |= ((byte >> j) & 1u) << (4 * j + 3);   // sign bit of nibble j
                o[q] = v;
            }
        }
        out = v4u{o[0], o[1], o[2], o[3]};
    }
    ((gptr<v4u>)(BITS == 8 ? F.desc8 : F.desc4))[(size_t)row * CHUNKS + chunk] = out;
}

// The descriptor rows in FeatureVector order (FrameDev::descfv).  thread = one 16-byte half of one row; the rows from the
// FeatureVector's length up to the padded end are zero.
__global__ __launch_bounds__(256) void expand_image_fv_kernel(FrameDev *__restrict__ table, const int32_t *__restrict__ ids, int blocks_per_frame) {
    const int f = blockIdx.x / blocks_per_frame;
    const FrameG F(table[ids[f]]);
    const int row = (blockIdx.x - f * blocks_per_frame) * 128 + (int)(threadIdx.x >> 1), half = threadIdx.x & 1;
    const int rows = ((F.n + 31) & ~31) + 32;
    if (row >= rows) return;
    const int nfv = F.n_nodes > 0 ? (int)F.fv_off[F.n_nodes] : 0;
    v4u out = {0, 0, 0, 0};
    if (row < nfv) out = ((gptr<const v4u>)F.desc)[2 * (size_t)F.fv_idx[row] + half];
    ((gptr<v4u>)F.descfv)[2 * (size_t)row + half] = out;
}

void orbm_drop_fv_image(orbm_ctx *ctx, int32_t id) {
    if (id < 0 || (size_t)id >= ctx->h_frames.size() || !ctx->h_frames[id].descfv) return;
    ctx->h_frames[id].descfv = nullptr;
    arena_unref(ctx, ctx->frame_arena4f[id]);
    ctx->frame_arena4f[id] = -1;
}

int orbm_ensure_image(orbm_ctx *ctx, int bits, int n, const int32_t *ids) {
    std::vector<int> &slot_arena = bits == 8 ? ctx->frame_arena8 : bits == 4 ? ctx->frame_arena4 : ctx->frame_arena4f;
    auto image_of = [bits](FrameDev &f) -> const uint32_t *& { return bits == 8 ? f.desc8 : bits == 4 ? f.desc4 : f.descfv; };
    const size_t row_bytes = bits == 8 ? 256 : bits == 4 ? 128 : 32;
    const int extra_rows = bits == 5 ? 32 : 0;         // descfv: a 32-row tile that starts at a node's last rows reads on
    std::vector<int32_t> need;
    if (ids) {
        for (int i = 0; i < n; i++) {
            const int32_t id = ids[i];
            if (image_of(ctx->h_frames[id]) || slot_arena[id] == -2) continue;
            slot_arena[id] = -2;                                           // listed (also de-duplicates)
            need.push_back(id);
        }
    } else {
        for (size_t id = 0; id < ctx->h_frames.size(); id++)
            if (ctx->frame_arena[id] >= 0 && !image_of(ctx->h_frames[id])) need.push_back((int32_t)id);
    }
    if (need.empty()) return ORBM_OK;
    for (int32_t id : need) slot_arena[id] = -1;
    size_t total = 0;
    int max_n8 = 32;
    std::vector<size_t> off(need.size());
    for (size_t k = 0; k < need.size(); k++) {
        const int n8 = std::max(32, (ctx->h_frames[need[k]].n + 31) & ~31) + extra_rows;
        off[k] = total; total += (size_t)n8 * row_bytes;
        max_n8 = std::max(max_n8, n8);
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // everything that can fail without touching the host mirror first: the id list's workspace, then the arena
    { int rc = orbm_ws_reserve(ctx, need.size() * 4); if (rc) return rc; }
    Arena ar;
    { int rc = orbm_arena_alloc(ctx, total, &ar); if (rc) return rc; }
    ar.live = (int)need.size();
    int arena_id = -1;
    for (size_t i = 0; i < ctx->arenas.size(); i++) if (!ctx->arenas[i].base) { arena_id = (int)i; break; }
    if (arena_id < 0) { ctx->arenas.push_back(ar); arena_id = (int)ctx->arenas.size() - 1; } else ctx->arenas[arena_id] = ar;
    size_t lo = SIZE_MAX, hi = 0;
    for (size_t k = 0; k < need.size(); k++) {
        const int32_t id = need[k];
        image_of(ctx->h_frames[id]) = (const uint32_t *)((char *)ar.base + off[k]);
        slot_arena[id] = arena_id;
        lo = std::min(lo, (size_t)id); hi = std::max(hi, (size_t)id);
    }
    // From here on the host mirror says "these frames have an image".  If a copy, the launch or the synchronisation fails the
    // images were not (or not provably) built: take the pointers back, on the host and -- best effort -- in the device table,
    // and release the arena, so that the next call builds them again instead of multiplying uninitialised rows.
    auto build = [&]() -> int {
        const size_t tab_bytes = (hi - lo + 1) * sizeof(FrameDev), id_bytes = need.size() * 4, o_ids = align_up(tab_bytes, 256);
        if (char *pin = (char *)orbm_pin_up(ctx, o_ids + id_bytes)) {
            memcpy(pin, ctx->h_frames.data() + lo, tab_bytes);
            memcpy(pin + o_ids, need.data(), id_bytes);
            HIPCHK(ctx, hipMemcpyAsync(ctx->d_frames + lo, pin, tab_bytes, hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(ctx, hipMemcpyAsync(ctx->ws, pin + o_ids, id_bytes, hipMemcpyHostToDevice, ctx->stream));
        } else {
            HIPCHK(ctx, hipMemcpy(ctx->d_frames + lo, ctx->h_frames.data() + lo, tab_bytes, hipMemcpyHostToDevice));
            HIPCHK(ctx, hipMemcpy(ctx->ws, need.data(), id_bytes, hipMemcpyHostToDevice));
        }
        if (bits == 5) {
            const int bpf = (max_n8 + 127) / 128;
            hipLaunchKernelGGL(expand_image_fv_kernel, dim3((unsigned)(need.size() * bpf)), dim3(256), 0, ctx->stream, ctx->d_frames, (const int32_t *)ctx->ws, bpf);
        } else if (bits == 8) {
            const int bpf = max_n8 / 16;
            hipLaunchKernelGGL(expand_image_kernel<8>, dim3((unsigned)(need.size() * bpf)), dim3(256), 0, ctx->stream, ctx->d_frames, (const int32_t *)ctx->ws, bpf);
        } else {
            const int bpf = max_n8 / 32;
            hipLaunchKernelGGL(expand_image_kernel<4>, dim3((unsigned)(need.size() * bpf)), dim3(256), 0, ctx->stream, ctx->d_frames, (const int32_t *)ctx->ws, bpf);
        }
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));    // the pinned block and the id list in ws are reused by the caller
        return ORBM_OK;
    };
    const int rc = build();
    if (rc) {
        const std::string why = ctx->err;                  // the message of the step that failed, not of the clean-up
        for (int32_t id : need) { image_of(ctx->h_frames[id]) = nullptr; slot_arena[id] = -1; }
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipMemcpy(ctx->d_frames + lo, ctx->h_frames.data() + lo, (hi - lo + 1) * sizeof(FrameDev), hipMemcpyHostToDevice);
        ctx->arenas[arena_id].live = 0;
        orbm_arena_release(ctx, ctx->arenas[arena_id]);
        ctx->err = why;
        return rc;
    }
    return ORBM_OK;
}
