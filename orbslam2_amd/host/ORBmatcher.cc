// ORBmatcher.cc -- host side of the drop-in ORB_SLAM2::ORBmatcher: flatten -> C ABI (HIP kernels) -> re-apply
// the pointer side effects.  No matching arithmetic happens here; if liborbmatch_hip.so cannot create a
// context the methods throw (there is no CPU fallback).
#include "ORBmatcher.h"

#include <atomic>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <tuple>

#include "../../include/orbmatch.h"
#include "host_ctx.h"
#include "host_gates.h"

using namespace std;

namespace ORB_SLAM2 {

const int ORBmatcher::TH_HIGH = 100;      // ORBmatcher.cc:37
const int ORBmatcher::TH_LOW = 50;        // ORBmatcher.cc:38
const int ORBmatcher::HISTO_LENGTH = 30;  // ORBmatcher.cc:39

// Frame's static members (Frame.cc:32-34) belong to the host project's Frame.cc; a stand-alone build that uses
// orbslam_types.h defines them in its own translation unit (tests/cpp/test_orbmatcher.cc does).

namespace {
std::atomic<int> g_device{-1};
std::mutex g_devices_mutex;
std::vector<int> g_devices;        // SetDevices: the GPUs (contexts) a batched SearchByBoW is cut over; empty = one
}

// ---- one C-ABI context per calling thread (the reference calls the matcher from three threads) -------------
// The context also owns the thread's RESIDENCY table: device copies of the Frames / KeyFrames this thread has searched,
// kept between calls (BASELINE north star: "descriptor storage, the grid-cell index and the FeatureVector grouping
// change to device-resident layouts").  Keys are (kind, mnId, descriptor buffer, N) + a content fingerprint, never the
// object's address, so a recycled Frame object (Tracking's mCurrentFrame) or a freed KeyFrame cannot alias.
namespace detail {
namespace {
struct ResKey {
    bool kf; long unsigned int id; const void *desc; int n;
    bool operator<(const ResKey &o) const { return std::tie(kf, id, desc, n) < std::tie(o.kf, o.id, o.desc, o.n); }
};
struct ResEntry {
    int32_t frame = -1;
    uint64_t print = 0;          // fingerprint of sampled descriptor rows and keypoints
    uint64_t content = 0;        // hash of EVERY descriptor row and keypoint, taken at upload
    uint64_t mp_version = ~0ull; // value of the caller's map-point version counter when `flags` was last refreshed
    size_t fv_size = 0;          // FeatureVector nodes the device copy was built from (0: none yet)
    vector<uint8_t> flags;       // mp_flags as last looked at
    bool flags_stale = false;    // `flags` is newer than the device copy (they went along with a search instead: see Acquire)
    std::list<ResKey>::iterator lru;
};
struct ThreadCtx {
    orbm_ctx *ctx = nullptr;
    int device = -1;             // -1: SetDevice / $ORBMATCH_DEVICE / 0
    std::map<ResKey, ResEntry> resident;
    std::list<ResKey> lru[2];    // [0] Frames, [1] KeyFrames; front = most recent
    int hold = 0;                // > 0 while a batched call collects handles: nothing is dropped until it ends
    ~ThreadCtx() { if (ctx) orbm_destroy(ctx); }   // frees every device copy with it
};
// A calling thread owns one context (and residency table) per GPU it uses: `primary` for everything, `extra[d - 1]` for entry
// d >= 1 of SetDevices.  A worker thread of a multi-GPU batched call works on one of its PARENT's states (tl_override).
struct ThreadStates {
    ThreadCtx primary;
    std::vector<std::unique_ptr<ThreadCtx>> extra;
};
thread_local ThreadCtx *tl_override = nullptr;
ThreadStates &States() {
    static thread_local ThreadStates s;
    return s;
}
ThreadCtx &Thread() { return tl_override ? *tl_override : States().primary; }
std::atomic<size_t> g_limit_kf{16384}, g_limit_frames{8};
std::atomic<int> g_verify{1};                                    // 0 key only, 1 sampled fingerprint (default), 2 full content hash per call
std::atomic<const std::atomic<uint64_t> *> g_mp_version{nullptr};
}  // namespace
orbm_ctx *NewCtxOn(int dev) {
    if (dev < 0) dev = g_device;
    if (dev < 0) { const char *e = getenv("ORBMATCH_DEVICE"); dev = e ? atoi(e) : 0; }
    orbm_ctx *c = orbm_create(dev);
    if (!c) throw runtime_error(string("ORBmatcher: cannot create a GPU context: ") + orbm_last_error(nullptr));
    return c;
}
orbm_ctx *NewCtx() { return NewCtxOn(-1); }
orbm_ctx *HostCtx() {
    ThreadCtx &t = Thread();
    if (!t.ctx) t.ctx = NewCtxOn(t.device);
    return t.ctx;
}
void CheckOn(orbm_ctx *ctx, int rc, const char *what) {
    if (rc < 0) throw runtime_error(string("ORBmatcher: ") + what + " failed: " + orbm_last_error(ctx));
}
void HostCheck(int rc, const char *what) { CheckOn(HostCtx(), rc, what); }
}  // namespace detail

namespace {
orbm_ctx *Ctx() { return detail::HostCtx(); }
void Check(int rc, const char *what) { detail::HostCheck(rc, what); }

// ---- flat snapshots -----------------------------------------------------------------------------------------
struct Flat {
    vector<float> x, y, angle, angle_raw;
    vector<int32_t> octave;
    vector<uint8_t> flags;
    vector<uint32_t> fv_node, fv_off, fv_idx;
    orbm_frame_view v;
};

uint8_t MpFlags(MapPoint *p) {
    if (!p) return 0;
    uint8_t f = ORBM_MP_PRESENT;
    if (p->isBad()) f |= ORBM_MP_BAD;
    if (p->Observations() > 0) f |= ORBM_MP_OBS;
    return f;
}

template <class FrameLike>
void FlattenCommon(Flat &o, const FrameLike &F, const vector<MapPoint *> &mps, bool with_fv) {
    const int n = F.N;
    o.x.resize(n); o.y.resize(n); o.angle.resize(n); o.angle_raw.resize(n); o.octave.resize(n); o.flags.assign(n, 0);
    for (int i = 0; i < n; i++) {
        o.x[i] = F.mvKeysUn[i].pt.x; o.y[i] = F.mvKeysUn[i].pt.y;
        o.angle[i] = F.mvKeysUn[i].angle; o.angle_raw[i] = F.mvKeys.empty() ? F.mvKeysUn[i].angle : F.mvKeys[i].angle;
        o.octave[i] = F.mvKeysUn[i].octave;
        if (i < (int)mps.size()) o.flags[i] = MpFlags(mps[i]);
    }
    o.fv_off.assign(1, 0);
    if (with_fv)
        for (const auto &kv : F.mFeatVec) {   // ordered map: ascending node id, indices in stored order
            o.fv_node.push_back(kv.first);
            for (unsigned idx : kv.second) o.fv_idx.push_back(idx);
            o.fv_off.push_back((uint32_t)o.fv_idx.size());
        }
    memset(&o.v, 0, sizeof o.v);
    o.v.n = n;
    o.v.desc = n ? F.mDescriptors.template ptr<uint8_t>() : nullptr;
    o.v.x = o.x.data(); o.v.y = o.y.data(); o.v.angle = o.angle.data(); o.v.angle_raw = o.angle_raw.data();
    o.v.octave = o.octave.data();
    o.v.uright = F.mvuRight.empty() ? nullptr : F.mvuRight.data();
    o.v.mp_flags = o.flags.data();
    o.v.n_levels = F.mnScaleLevels;
    o.v.scale_factors = F.mvScaleFactors.data(); o.v.level_sigma2 = F.mvLevelSigma2.data();
    o.v.n_nodes = (int)o.fv_node.size();
    o.v.fv_node = o.fv_node.data(); o.v.fv_off = o.fv_off.data(); o.v.fv_idx = o.fv_idx.data();
}

void Flatten(Flat &o, const Frame &F, bool with_fv = false) {
    FlattenCommon(o, F, F.mvpMapPoints, with_fv);
    o.v.min_x = Frame::mnMinX; o.v.min_y = Frame::mnMinY; o.v.max_x = Frame::mnMaxX; o.v.max_y = Frame::mnMaxY;
    o.v.inv_w = Frame::mfGridElementWidthInv; o.v.inv_h = Frame::mfGridElementHeightInv;
}

void Flatten(Flat &o, KeyFrame *pKF, bool with_fv = false) {
    FlattenCommon(o, *pKF, pKF->GetMapPointMatches(), with_fv);
    o.v.min_x = (float)pKF->mnMinX; o.v.min_y = (float)pKF->mnMinY; o.v.max_x = (float)pKF->mnMaxX; o.v.max_y = (float)pKF->mnMaxY;
    o.v.inv_w = pKF->mfGridElementWidthInv; o.v.inv_h = pKF->mfGridElementHeightInv;
}

// ---- residency: the device copy of a Frame / KeyFrame, uploaded on first use by this thread and kept -----------------
template <class FrameLike>
uint64_t Fingerprint(const FrameLike &F) {
    uint64_t h = 1469598103934665603ull ^ (uint64_t)F.N;
    auto mix = [&h](const void *p, size_t n) {
        const uint8_t *b = static_cast<const uint8_t *>(p);
        for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
    };
    if (F.N > 0)
        for (int i : {0, F.N / 2, F.N - 1}) {
            mix(F.mDescriptors.template ptr<uint8_t>(i), 32);
            mix(&F.mvKeysUn[i].pt.x, sizeof(float)); mix(&F.mvKeysUn[i].pt.y, sizeof(float));
            mix(&F.mvKeysUn[i].octave, sizeof(int));
        }
    return h;
}

// every descriptor row and every keypoint field the device copy holds (x, y, angle, octave), 8 bytes at a time
template <class FrameLike>
uint64_t ContentHash(const FrameLike &F) {
    uint64_t h[4] = {0x9E3779B97F4A7C15ull ^ (uint64_t)F.N, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull, 0x27D4EB2F165667C5ull};
    auto mix = [&h](int lane, uint64_t w) { h[lane] = (h[lane] ^ w) * 0x100000001B3ull; h[lane] ^= h[lane] >> 29; };
    for (int i = 0; i < F.N; i++) {
        uint64_t w[4];
        memcpy(w, F.mDescriptors.template ptr<uint8_t>(i), 32);
        for (int k = 0; k < 4; k++) mix(k, w[k]);
        uint64_t a, b;
        const float xy[2] = {F.mvKeysUn[i].pt.x, F.mvKeysUn[i].pt.y};
        const float ao[2] = {F.mvKeysUn[i].angle, (float)F.mvKeysUn[i].octave};
        memcpy(&a, xy, 8); memcpy(&b, ao, 8);
        mix(i & 3, a); mix((i + 1) & 3, b);
    }
    return (h[0] * 31 + h[1]) ^ (h[2] * 131 + h[3]);
}

void DropResident(detail::ThreadCtx &t, std::map<detail::ResKey, detail::ResEntry>::iterator it) {
    orbm_frames_free(t.ctx, 1, &it->second.frame);
    t.lru[it->first.kf].erase(it->second.lru);
    t.resident.erase(it);
}

// least recently used copies beyond the limits go (never the two most recent of a kind: one call needs at most two)
void TrimResident(detail::ThreadCtx &t) {
    for (int kf = 0; kf < 2; kf++) {
        const size_t limit = std::max<size_t>(4, kf ? detail::g_limit_kf.load() : detail::g_limit_frames.load());
        while (t.lru[kf].size() > limit) DropResident(t, t.resident.find(t.lru[kf].back()));
    }
}
struct HoldResident {             // scope of a batched call
    detail::ThreadCtx &t;
    HoldResident() : t(detail::Thread()) { t.hold++; }
    ~HoldResident() { if (--t.hold == 0 && t.ctx) TrimResident(t); }
};

template <class FrameLike>
detail::ResKey KeyOf(bool kf, const FrameLike &F) {
    return detail::ResKey{kf, F.mnId, F.N ? (const void *)F.mDescriptors.template ptr<uint8_t>() : nullptr, F.N};
}

// id of the device copy in this thread's context; mp_flags brought up to date (they change between calls, the rest does
// not: Frame.h:235-291 members are written once, by the constructor and ComputeBoW)
template <class FrameLike, class Source>
int32_t Acquire(bool kf, const FrameLike &F, Source &src, const vector<MapPoint *> &mps, bool with_fv, const uint8_t **defer = nullptr) {
    // defer (windowed searches into one frame, whose slot state changes between any two calls of a tracking thread): changed
    // flags are not sent to the device copy in a call of their own (a copy and a synchronisation) but handed back, to travel
    // with the search as its slot_flags; *defer stays null when the device copy is current
    if (defer) *defer = nullptr;
    orbm_ctx *ctx = Ctx();
    detail::ThreadCtx &t = detail::Thread();
    const detail::ResKey key = KeyOf(kf, F);
    const int verify = detail::g_verify.load(std::memory_order_relaxed);
    const uint64_t fp = verify >= 1 ? Fingerprint(F) : 0;
    auto it = t.resident.find(key);
    if (it != t.resident.end() && ((verify >= 1 && it->second.print != fp) || (verify >= 2 && it->second.content != ContentHash(F)) ||
                                   (with_fv && it->second.fv_size != F.mFeatVec.size()))) {
        DropResident(t, it);          // other content under the same key, or ComputeBoW ran after the upload
        it = t.resident.end();
    }
    if (it == t.resident.end()) {
        Flat f;
        Flatten(f, src, !F.mFeatVec.empty());
        // the flags of the upload come from the CALLER's snapshot of the map points (the one it maps the results through),
        // not from a second GetMapPointMatches(): LocalMapping may have changed the keyframe in between
        for (int i = 0; i < F.N; i++) f.flags[i] = i < (int)mps.size() ? MpFlags(mps[i]) : 0;
        const std::atomic<uint64_t> *vsrc = detail::g_mp_version.load(std::memory_order_acquire);
        detail::ResEntry e;
        if (vsrc) e.mp_version = vsrc->load(std::memory_order_acquire);
        Check(orbm_frames_upload(ctx, 1, &f.v, &e.frame), "orbm_frames_upload");
        e.print = verify >= 1 ? fp : Fingerprint(F); e.content = ContentHash(F); e.fv_size = F.mFeatVec.size(); e.flags.swap(f.flags);
        t.lru[kf].push_front(key);
        e.lru = t.lru[kf].begin();
        it = t.resident.emplace(key, std::move(e)).first;
        const int32_t id = it->second.frame;
        if (t.hold == 0) TrimResident(t);
        return id;
    }
    detail::ResEntry &e = it->second;
    t.lru[kf].splice(t.lru[kf].begin(), t.lru[kf], e.lru);
    // extension hook (SetMapPointVersion): the host project counts every change of a map point's bad flag / observations and of
    // a frame's or keyframe's map-point vector; while the counter stands where it stood at the last refresh the flags on the
    // device are current and the O(N) sweep below (a mutex-guarded isBad() + Observations() per keypoint) is skipped
    const std::atomic<uint64_t> *vsrc = detail::g_mp_version.load(std::memory_order_acquire);
    uint64_t vnow = 0;
    const int n = F.N;
    bool same = true;
    if (vsrc) vnow = vsrc->load(std::memory_order_acquire);
    if (!vsrc || vnow != e.mp_version) {
        for (int i = 0; i < n; i++) {
            const uint8_t fl = i < (int)mps.size() ? MpFlags(mps[i]) : 0;
            if (fl != e.flags[i]) { e.flags[i] = fl; same = false; }
        }
        if (vsrc) e.mp_version = vnow;  // read BEFORE the sweep: a change during it bumps the counter past this value
    }
    if ((!same || e.flags_stale) && n > 0) {
        if (defer) { *defer = e.flags.data(); e.flags_stale = true; }
        else { Check(orbm_frame_set_mp_flags(ctx, e.frame, e.flags.data()), "orbm_frame_set_mp_flags"); e.flags_stale = false; }
    }
    return e.frame;
}

int32_t Resident(const Frame &F, bool with_fv = false) { return Acquire(false, F, F, F.mvpMapPoints, with_fv); }
int32_t ResidentDeferFlags(const Frame &F, const uint8_t *&slot_flags) { return Acquire(false, F, F, F.mvpMapPoints, false, &slot_flags); }
int32_t Resident(KeyFrame *pKF, bool with_fv = false) { return Acquire(true, *pKF, pKF, pKF->GetMapPointMatches(), with_fv); }
int32_t Resident(KeyFrame *pKF, const vector<MapPoint *> &mps, bool with_fv) { return Acquire(true, *pKF, pKF, mps, with_fv); }

struct WinQ {
    vector<uint8_t> desc, flags;
    vector<float> u, v, radius, ur, angle;
    vector<int32_t> minL, maxL;
    explicit WinQ(int n) : desc((size_t)n * 32, 0), flags(n, 0), u(n, 0), v(n, 0), radius(n, 0), ur(n, 0), angle(n, 0), minL(n, 0), maxL(n, 0) {}
    void SetDesc(int i, const cv::Mat &d) { memcpy(&desc[(size_t)i * 32], d.ptr<uint8_t>(), 32); }
    orbm_win_queries View(bool with_ur, bool with_angle) const {
        orbm_win_queries q;
        memset(&q, 0, sizeof q);
        q.n = (int)flags.size();
        q.desc = desc.data(); q.u = u.data(); q.v = v.data(); q.radius = radius.data();
        q.min_level = minL.data(); q.max_level = maxL.data();
        q.ur = with_ur ? ur.data() : nullptr; q.angle = with_angle ? angle.data() : nullptr; q.flags = flags.data();
        return q;
    }
    // runs orbm_search_window_batch for one instance; returns nmatches, owner[] sized n_slots
    int Run(int mode, int32_t frame_id, int n_slots, int dist_th, bool check_ori, bool with_ur, bool with_angle,
            const uint8_t *slot_flags, vector<int32_t> &owner) {
        orbm_win_queries q = View(with_ur, with_angle);
        int32_t count = 0;
        owner.assign(max(n_slots, 1), -1);
        // one call: the query arrays travel with the search (one pinned copy in, one out, one synchronisation)
        Check(orbm_search_window_once(Ctx(), mode, frame_id, &q, dist_th, check_ori ? 1 : 0, slot_flags, owner.data(), &count),
              "orbm_search_window_once");
        return count;
    }
};

// claim-free best-1 (orbm_search_window_free_batch) of several (frame, query set) instances in one launch;
// best[k][i] = keypoint of frame k matched to query i, or -1
void RunFree(const vector<const WinQ *> &qs, const vector<int32_t> &frame_ids, int dist_th, bool chi2, vector<vector<int32_t>> &best) {
    const int K = (int)qs.size();
    vector<orbm_win_queries> views(K);
    int stride = 1;
    for (int k = 0; k < K; k++) { views[k] = qs[k]->View(true, false); stride = max(stride, views[k].n); }
    vector<int32_t> flat((size_t)K * stride, -1);
    // one call: the query sets travel with the search (one pinned copy in, one out, one synchronisation)
    Check(orbm_search_window_free_once(Ctx(), K, frame_ids.data(), views.data(), dist_th, chi2 ? 1 : 0, stride, flat.data()),
          "orbm_search_window_free_once");
    best.resize(K);
    for (int k = 0; k < K; k++) best[k].assign(flat.begin() + (size_t)k * stride, flat.begin() + (size_t)k * stride + views[k].n);
}

// Scw -> Rcw, tcw, Ow (ORBmatcher.cc:411-415 / 1313-1317).  cv::Mat / scalar is a convertTo with alpha = 1/scw; for
// CV_32F -> CV_32F OpenCV applies alpha in float.
void DecomposeSim3(const cv::Mat &Scw, float *Rcw, float *tcw, float *Ow) {
    const float *S = Scw.ptr<float>();
    const float row0[3] = {S[0], S[1], S[2]};
    const float scw = sqrt(orbm_host::dot3(row0, row0));
    const float inv = (float)(1.0 / (double)scw);
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) Rcw[3 * r + c] = S[4 * r + c] * inv;
        tcw[r] = S[4 * r + 3] * inv;
    }
    orbm_host::gemm3(Rcw, 3, true, tcw, -1.0, nullptr, Ow);
}

}  // namespace

void ORBmatcher::SetDevice(int device) { g_device = device; }
void ORBmatcher::SetResidentLimits(size_t nKeyFrames, size_t nFrames) { detail::g_limit_kf = nKeyFrames; detail::g_limit_frames = nFrames; }
void ORBmatcher::SetResidencyVerify(int level) { detail::g_verify = level < 0 ? 0 : level > 2 ? 2 : level; }
void ORBmatcher::SetMapPointVersion(const std::atomic<uint64_t> *counter) { detail::g_mp_version.store(counter, std::memory_order_release); }
void ORBmatcher::Prefetch(KeyFrame *pKF) { if (pKF && pKF->N > 0) Resident(pKF, true); }
namespace {
template <class Fn>
void ForEachState(Fn &&fn) {                       // every context of the calling thread
    detail::ThreadStates &s = detail::States();
    fn(s.primary);
    for (auto &e : s.extra) if (e) fn(*e);
}
}  // namespace
void ORBmatcher::Evict(KeyFrame *pKF) {
    if (!pKF) return;
    ForEachState([&](detail::ThreadCtx &t) {
        if (!t.ctx) return;
        auto it = t.resident.find(KeyOf(true, *pKF));
        if (it != t.resident.end()) DropResident(t, it);
    });
}
void ORBmatcher::Evict(const Frame &F) {
    ForEachState([&](detail::ThreadCtx &t) {
        if (!t.ctx) return;
        auto it = t.resident.find(KeyOf(false, F));
        if (it != t.resident.end()) DropResident(t, it);
    });
}
void ORBmatcher::SetDevices(const std::vector<int> &devices) {
    std::lock_guard<std::mutex> lock(g_devices_mutex);
    g_devices = devices;
}

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

int ORBmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b) {
    const uint64_t *pa = a.ptr<uint64_t>(), *pb = b.ptr<uint64_t>();
    return __builtin_popcountll(pa[0] ^ pb[0]) + __builtin_popcountll(pa[1] ^ pb[1]) +
           __builtin_popcountll(pa[2] ^ pb[2]) + __builtin_popcountll(pa[3] ^ pb[3]);
}

// ---- ORBmatcher.cc:57-166 ---------------------------------------------------------------------------------------
int ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint *> &vpMapPoints, const float th) {
    const int Q = (int)vpMapPoints.size();
    if (Q == 0 || F.N == 0) return 0;
    vector<uint8_t> desc((size_t)Q * 32, 0), flags(Q, 0);
    vector<float> px(Q, 0), py(Q, 0), pxr(Q, 0), vc(Q, 0);
    vector<int32_t> lvl(Q, 0);
    for (int i = 0; i < Q; i++) {
        MapPoint *pMP = vpMapPoints[i];
        if (!pMP->mbTrackInView) continue;       // :69
        if (pMP->isBad()) continue;              // :72
        flags[i] = ORBM_Q_VALID | (pMP->Observations() > 0 ? ORBM_Q_OBS : 0);
        px[i] = pMP->mTrackProjX; py[i] = pMP->mTrackProjY; pxr[i] = pMP->mTrackProjXR;
        lvl[i] = pMP->mnTrackScaleLevel; vc[i] = pMP->mTrackViewCos;
        memcpy(&desc[(size_t)i * 32], pMP->GetDescriptor().ptr<uint8_t>(), 32);
    }
    const uint8_t *slot_flags = nullptr;
    const int32_t fid = ResidentDeferFlags(F, slot_flags);
    orbm_proj_queries q;
    q.n = Q; q.desc = desc.data(); q.proj_x = px.data(); q.proj_y = py.data(); q.proj_xr = pxr.data();
    q.level = lvl.data(); q.view_cos = vc.data(); q.flags = flags.data();
    int32_t count = 0;
    vector<int32_t> owner(F.N, -1);
    Check(orbm_search_proj_points_once(Ctx(), fid, &q, th, mfNNratio, slot_flags, owner.data(), &count), "orbm_search_proj_points_once");
    for (int idx = 0; idx < F.N; idx++)
        if (owner[idx] >= 0) F.mvpMapPoints[idx] = vpMapPoints[owner[idx]];   // :160
    return count;
}

// ---- ORBmatcher.cc:218-385 ----------------------------------------------------------------------------------------
int ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint *> &vpMapPointMatches) {
    const vector<MapPoint *> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = vector<MapPoint *>(F.N, static_cast<MapPoint *>(NULL));   // :225
    if (F.N == 0 || pKF->N == 0) return 0;
    const int32_t ida = Resident(pKF, vpMapPointsKF, true), idb = Resident(F, true);
    vector<int32_t> match(F.N, -1);
    int32_t count = 0;
    Check(orbm_search_bow_batch(Ctx(), 0, 1, &ida, &idb, mfNNratio, mbCheckOrientation ? 1 : 0, F.N, match.data(), &count, 0),
          "orbm_search_bow_batch");
    for (int i = 0; i < F.N; i++)
        if (match[i] >= 0) vpMapPointMatches[i] = vpMapPointsKF[match[i]];          // :312
    return count;
}

// ---- ORBmatcher.cc:703-854 ----------------------------------------------------------------------------------------
int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint *> &vpMatches12) {
    const vector<MapPoint *> vpMapPoints1 = pKF1->GetMapPointMatches();
    const vector<MapPoint *> vpMapPoints2 = pKF2->GetMapPointMatches();
    vpMatches12 = vector<MapPoint *>(vpMapPoints1.size(), static_cast<MapPoint *>(NULL));   // :715
    if (pKF1->N == 0 || pKF2->N == 0) return 0;
    const int32_t ida = Resident(pKF1, vpMapPoints1, true), idb = Resident(pKF2, vpMapPoints2, true);
    vector<int32_t> match(pKF1->N, -1);
    int32_t count = 0;
    Check(orbm_search_bow_batch(Ctx(), 1, 1, &ida, &idb, mfNNratio, mbCheckOrientation ? 1 : 0, pKF1->N, match.data(), &count, 0),
          "orbm_search_bow_batch");
    for (int i = 0; i < pKF1->N; i++)
        if (match[i] >= 0) vpMatches12[i] = vpMapPoints2[match[i]];                  // :798
    return count;
}

// ---- the candidate loops (Tracking.cc:1778-1801, LoopClosing.cc:433-477) in one launch ---------------------------------
namespace {
// Candidates [k0, k1) of a batched SearchByBoW on the CURRENT context (detail::Thread()): variant 0 = (candidate KF, Frame F),
// variant 1 = (pKF1, candidate KF).  Fills nmatches[k] and out[k] for its candidates only.
template <class Anchor>
void BowSlice(int variant, Anchor &anchor, int anchorN, const vector<MapPoint *> *anchorMps, const vector<KeyFrame *> &cands, int k0, int k1,
              float ratio, bool checkOri, size_t outSize, vector<int> &nmatches, vector<vector<MapPoint *>> &out) {
    HoldResident hold;
    vector<int32_t> ida, idb, which;
    vector<vector<MapPoint *>> mps(k1 - k0);
    int32_t anchorId = -1;
    if (anchorN > 0) {
        if constexpr (std::is_same<Anchor, Frame>::value) anchorId = Resident(anchor, true);
        else anchorId = Resident(anchor, *anchorMps, true);
    }
    for (int k = k0; k < k1; k++) {
        KeyFrame *pKF = cands[k];
        if (!pKF || pKF->isBad()) continue;                                        // Tracking.cc:1781-1782 / LoopClosing.cc:441-445
        mps[k - k0] = pKF->GetMapPointMatches();
        out[k] = vector<MapPoint *>(outSize, static_cast<MapPoint *>(NULL));
        if (anchorN == 0 || pKF->N == 0) continue;
        const int32_t cid = Resident(pKF, mps[k - k0], true);
        if (variant == 0) { ida.push_back(cid); idb.push_back(anchorId); }
        else { ida.push_back(anchorId); idb.push_back(cid); }
        which.push_back(k);
    }
    const int P = (int)which.size();
    if (P == 0) return;
    vector<int32_t> match((size_t)P * anchorN, -1), count(P, 0);
    Check(orbm_search_bow_batch(Ctx(), variant, P, ida.data(), idb.data(), ratio, checkOri ? 1 : 0, anchorN, match.data(), count.data(), 0),
          "orbm_search_bow_batch");
    for (int p = 0; p < P; p++) {
        const int k = which[p];
        const int32_t *m = &match[(size_t)p * anchorN];
        for (int i = 0; i < anchorN; i++)
            if (m[i] >= 0) out[k][i] = mps[k - k0][m[i]];
        nmatches[k] = count[p];
    }
}

// The whole candidate list: on the calling thread's context, or -- after SetDevices({...}) with more than one entry -- cut into
// contiguous slices, one per listed GPU, each slice on a worker thread with the calling thread's context (and residency table)
// for that GPU.  Pairs are independent, so the result is the single-context result (LoopClosing.cc:433-477 over shards).
template <class Anchor>
void BowBatch(int variant, Anchor &anchor, int anchorN, const vector<MapPoint *> *anchorMps, const vector<KeyFrame *> &cands, float ratio,
              bool checkOri, size_t outSize, vector<int> &nmatches, vector<vector<MapPoint *>> &out) {
    const int K = (int)cands.size();
    vector<int> devs;
    { std::lock_guard<std::mutex> lock(g_devices_mutex); devs = g_devices; }
    const int D = std::min((int)devs.size(), K);
    if (D <= 1 || detail::tl_override) {
        BowSlice(variant, anchor, anchorN, anchorMps, cands, 0, K, ratio, checkOri, outSize, nmatches, out);
        return;
    }
    detail::ThreadStates &st = detail::States();
    if (st.primary.device < 0 && !st.primary.ctx) st.primary.device = devs[0];
    while ((int)st.extra.size() < D - 1) st.extra.emplace_back();
    vector<detail::ThreadCtx *> state(D);
    state[0] = &st.primary;
    for (int d = 1; d < D; d++) {
        auto &e = st.extra[d - 1];
        if (!e || (e->device != devs[d])) { e.reset(new detail::ThreadCtx()); e->device = devs[d]; }
        state[d] = e.get();
    }
    vector<std::exception_ptr> err(D);
    auto run = [&](int d) {
        try {
            BowSlice(variant, anchor, anchorN, anchorMps, cands, (int)((long long)K * d / D), (int)((long long)K * (d + 1) / D), ratio, checkOri,
                     outSize, nmatches, out);
        } catch (...) { err[d] = std::current_exception(); }
    };
    vector<std::thread> workers;
    for (int d = 1; d < D; d++)
        workers.emplace_back([&, d] { detail::tl_override = state[d]; run(d); detail::tl_override = nullptr; });
    run(0);
    for (auto &w : workers) w.join();
    for (int d = 0; d < D; d++) if (err[d]) std::rethrow_exception(err[d]);
}
}  // namespace

vector<int> ORBmatcher::SearchByBoW(const vector<KeyFrame *> &vpKFs, Frame &F, vector<vector<MapPoint *>> &vvpMapPointMatches) {
    const int K = (int)vpKFs.size();
    vector<int> nmatches(K, 0);
    vvpMapPointMatches.assign(K, vector<MapPoint *>());
    BowBatch(0, F, F.N, nullptr, vpKFs, mfNNratio, mbCheckOrientation, (size_t)F.N, nmatches, vvpMapPointMatches);
    return nmatches;
}

vector<int> ORBmatcher::SearchByBoW(KeyFrame *pKF1, const vector<KeyFrame *> &vpKF2s, vector<vector<MapPoint *>> &vvpMatches12) {
    const int K = (int)vpKF2s.size();
    vector<int> nmatches(K, 0);
    vvpMatches12.assign(K, vector<MapPoint *>());
    const vector<MapPoint *> vpMapPoints1 = pKF1->GetMapPointMatches();
    BowBatch(1, pKF1, pKF1->N, &vpMapPoints1, vpKF2s, mfNNratio, mbCheckOrientation, vpMapPoints1.size(), nmatches, vvpMatches12);
    return nmatches;
}

// ---- ORBmatcher.cc:562-689 ----------------------------------------------------------------------------------------
int ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, vector<cv::Point2f> &vbPrevMatched, vector<int> &vnMatches12, int windowSize) {
    const int n1 = (int)F1.mvKeysUn.size();
    vnMatches12 = vector<int>(n1, -1);            // :566
    if (n1 == 0 || F2.N == 0) return 0;
    const int32_t ida = Resident(F1), idb = Resident(F2);
    vector<float> prev((size_t)n1 * 2);
    for (int i = 0; i < n1; i++) { prev[2 * i] = vbPrevMatched[i].x; prev[2 * i + 1] = vbPrevMatched[i].y; }
    vector<int32_t> m12(n1, -1);
    int n = orbm_search_init(Ctx(), ida, idb, prev.data(), windowSize, mfNNratio, mbCheckOrientation ? 1 : 0, m12.data());
    Check(n, "orbm_search_init");
    for (int i = 0; i < n1; i++) {
        vnMatches12[i] = m12[i];
        vbPrevMatched[i] = cv::Point2f(prev[2 * i], prev[2 * i + 1]);   // :684-686 (unchanged where unmatched)
    }
    return n;
}

// ---- ORBmatcher.cc:868-1093 ---------------------------------------------------------------------------------------
int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, vector<pair<size_t, size_t>> &vMatchedPairs, const bool bOnlyStereo) {
    // epipole of camera 1 in image 2 (:876-884)
    cv::Mat Cw = pKF1->GetCameraCenter(), R2w = pKF2->GetRotation(), t2w = pKF2->GetTranslation();
    float C2[3];
    orbm_host::gemm3(R2w.ptr<float>(), 3, false, Cw.ptr<float>(), 1.0, t2w.ptr<float>(), C2);
    const float invz = 1.0f / C2[2];
    const float epi[2] = {pKF2->fx * C2[0] * invz + pKF2->cx, pKF2->fy * C2[1] * invz + pKF2->cy};
    vMatchedPairs.clear();
    if (pKF1->N == 0 || pKF2->N == 0) return 0;
    const int32_t ida = Resident(pKF1, true), idb = Resident(pKF2, true);
    float f12[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) f12[3 * r + c] = F12.at<float>(r, c);
    vector<int32_t> m12(pKF1->N, -1);
    int32_t count = 0;
    Check(orbm_search_triang_batch(Ctx(), 1, &ida, &idb, f12, epi, bOnlyStereo ? 1 : 0, mbCheckOrientation ? 1 : 0,
                                   pKF1->N, m12.data(), &count, 0), "orbm_search_triang_batch");
    vMatchedPairs.reserve(count);
    for (size_t i = 0; i < m12.size(); i++)                                           // :1085-1090
        if (m12[i] >= 0) vMatchedPairs.push_back(make_pair(i, (size_t)m12[i]));
    return count;
}

// ---- ORBmatcher.cc:1768-1950 --------------------------------------------------------------------------------------
int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono) {
    if (CurrentFrame.N == 0 || LastFrame.N == 0) return 0;
    const float *Tcw = CurrentFrame.mTcw.ptr<float>();   // 4x4 row-major
    const float tcw[3] = {Tcw[3], Tcw[7], Tcw[11]};
    float twc[3];
    orbm_host::gemm3(Tcw, 4, true, tcw, -1.0, nullptr, twc);                           // :1783
    const float *Tlw = LastFrame.mTcw.ptr<float>();
    const float tlw[3] = {Tlw[3], Tlw[7], Tlw[11]};
    float tlc[3];
    orbm_host::gemm3(Tlw, 4, false, twc, 1.0, tlw, tlc);                               // :1791
    const bool bForward = tlc[2] > CurrentFrame.mb && !bMono;                          // :1795
    const bool bBackward = -tlc[2] > CurrentFrame.mb && !bMono;                        // :1796

    WinQ q(LastFrame.N);
    for (int i = 0; i < LastFrame.N; i++) {
        MapPoint *pMP = LastFrame.mvpMapPoints[i];
        if (!pMP) continue;
        if (LastFrame.mvbOutlier[i]) continue;
        cv::Mat x3Dw = pMP->GetWorldPos();
        float x3Dc[3];
        orbm_host::gemm3(Tcw, 4, false, x3Dw.ptr<float>(), 1.0, tcw, x3Dc);            // :1813
        const float xc = x3Dc[0], yc = x3Dc[1];
        const float invzc = 1.0 / x3Dc[2];
        if (invzc < 0) continue;
        float u = Frame::fx * xc * invzc + Frame::cx;
        float v = Frame::fy * yc * invzc + Frame::cy;
        if (u < Frame::mnMinX || u > Frame::mnMaxX) continue;
        if (v < Frame::mnMinY || v > Frame::mnMaxY) continue;
        const int nLastOctave = LastFrame.mvKeys[i].octave;
        q.radius[i] = th * CurrentFrame.mvScaleFactors[nLastOctave];                   // :1845
        if (bForward) { q.minL[i] = nLastOctave; q.maxL[i] = -1; }                     // :1859-1864
        else if (bBackward) { q.minL[i] = 0; q.maxL[i] = nLastOctave; }
        else { q.minL[i] = nLastOctave - 1; q.maxL[i] = nLastOctave + 1; }
        q.u[i] = u; q.v[i] = v;
        q.ur[i] = u - CurrentFrame.mbf * invzc;                                        // :1887
        q.angle[i] = LastFrame.mvKeysUn[i].angle;
        q.flags[i] = ORBM_Q_VALID | (pMP->Observations() > 0 ? ORBM_Q_OBS : 0);
        q.SetDesc(i, pMP->GetDescriptor());
    }
    const uint8_t *slot_flags = nullptr;
    const int32_t fid = ResidentDeferFlags(CurrentFrame, slot_flags);
    vector<int32_t> owner;
    const int n = q.Run(0, fid, CurrentFrame.N, TH_HIGH, mbCheckOrientation, true, true, slot_flags, owner);
    for (int idx = 0; idx < CurrentFrame.N; idx++) {
        if (owner[idx] >= 0) CurrentFrame.mvpMapPoints[idx] = LastFrame.mvpMapPoints[owner[idx]];   // :1906
        else if (owner[idx] == -2) CurrentFrame.mvpMapPoints[idx] = static_cast<MapPoint *>(NULL);  // :1942
    }
    return n;
}

// ---- ORBmatcher.cc:1963-2123 --------------------------------------------------------------------------------------
int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint *> &sAlreadyFound, const float th, const int ORBdist) {
    if (CurrentFrame.N == 0) return 0;
    const float *Tcw = CurrentFrame.mTcw.ptr<float>();
    const float tcw[3] = {Tcw[3], Tcw[7], Tcw[11]};
    float Ow[3];
    orbm_host::gemm3(Tcw, 4, true, tcw, -1.0, nullptr, Ow);                            // :1969
    const vector<MapPoint *> vpMPs = pKF->GetMapPointMatches();
    WinQ q((int)vpMPs.size());
    for (size_t i = 0; i < vpMPs.size(); i++) {
        MapPoint *pMP = vpMPs[i];
        if (!pMP) continue;
        if (pMP->isBad() || sAlreadyFound.count(pMP)) continue;
        cv::Mat x3Dw = pMP->GetWorldPos();
        float x3Dc[3];
        orbm_host::gemm3(Tcw, 4, false, x3Dw.ptr<float>(), 1.0, tcw, x3Dc);
        const float xc = x3Dc[0], yc = x3Dc[1];
        const float invzc = 1.0 / x3Dc[2];
        const float u = Frame::fx * xc * invzc + Frame::cx;
        const float v = Frame::fy * yc * invzc + Frame::cy;
        if (u < Frame::mnMinX || u > Frame::mnMaxX) continue;
        if (v < Frame::mnMinY || v > Frame::mnMaxY) continue;
        const float *pw = x3Dw.ptr<float>();
        const float PO[3] = {pw[0] - Ow[0], pw[1] - Ow[1], pw[2] - Ow[2]};
        float dist3D = orbm_host::norm3(PO);
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const int nPredictedLevel = pMP->PredictScale(dist3D, &CurrentFrame);
        q.radius[i] = th * CurrentFrame.mvScaleFactors[nPredictedLevel];                // :2026
        q.minL[i] = nPredictedLevel - 1; q.maxL[i] = nPredictedLevel + 1;               // :2030
        q.u[i] = u; q.v[i] = v;
        q.angle[i] = pKF->mvKeysUn[i].angle;
        q.flags[i] = ORBM_Q_VALID | ORBM_Q_OBS;
        q.SetDesc((int)i, pMP->GetDescriptor());
    }
    const uint8_t *slot_flags = nullptr;
    const int32_t fid = ResidentDeferFlags(CurrentFrame, slot_flags);
    vector<int32_t> owner;
    const int n = q.Run(1, fid, CurrentFrame.N, ORBdist, mbCheckOrientation, false, true, slot_flags, owner);
    for (int idx = 0; idx < CurrentFrame.N; idx++) {
        if (owner[idx] >= 0) CurrentFrame.mvpMapPoints[idx] = vpMPs[owner[idx]];        // :2066
        else if (owner[idx] == -2) CurrentFrame.mvpMapPoints[idx] = NULL;               // :2115
    }
    return n;
}

// ---- ORBmatcher.cc:401-542 ----------------------------------------------------------------------------------------
int ORBmatcher::SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint *> &vpPoints, vector<MapPoint *> &vpMatched, int th) {
    if (pKF->N == 0 || vpPoints.empty()) return 0;
    const float &fx = pKF->fx, &fy = pKF->fy, &cx = pKF->cx, &cy = pKF->cy;
    float Rcw[9], tcw[3], Ow[3];
    DecomposeSim3(Scw, Rcw, tcw, Ow);                                                     // :411-415
    set<MapPoint *> spAlreadyFound(vpMatched.begin(), vpMatched.end());
    spAlreadyFound.erase(static_cast<MapPoint *>(NULL));

    const int Q = (int)vpPoints.size();
    WinQ q(Q);
    for (int iMP = 0; iMP < Q; iMP++) {
        MapPoint *pMP = vpPoints[iMP];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        cv::Mat p3Dw = pMP->GetWorldPos();
        float p3Dc[3];
        orbm_host::gemm3(Rcw, 3, false, p3Dw.ptr<float>(), 1.0, tcw, p3Dc);
        if (p3Dc[2] < 0.0) continue;
        const float invz = 1 / p3Dc[2];
        const float x = p3Dc[0] * invz, y = p3Dc[1] * invz;
        const float u = fx * x + cx, v = fy * y + cy;
        if (!pKF->IsInImage(u, v)) continue;
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        const float *pw = p3Dw.ptr<float>();
        const float PO[3] = {pw[0] - Ow[0], pw[1] - Ow[1], pw[2] - Ow[2]};
        const float dist = orbm_host::norm3(PO);
        if (dist < minDistance || dist > maxDistance) continue;
        cv::Mat Pn = pMP->GetNormal();
        if (orbm_host::dot3(PO, Pn.ptr<float>()) < 0.5 * dist) continue;                  // :487
        const int nPredictedLevel = pMP->PredictScale(dist, pKF);
        q.radius[iMP] = th * pKF->mvScaleFactors[nPredictedLevel];                        // :495
        q.minL[iMP] = nPredictedLevel - 1; q.maxL[iMP] = nPredictedLevel;                 // :517
        q.u[iMP] = u; q.v[iMP] = v;
        q.flags[iMP] = ORBM_Q_VALID | ORBM_Q_OBS;
        q.SetDesc(iMP, pMP->GetDescriptor());
    }
    const int32_t fid = Resident(pKF);
    vector<uint8_t> slot(pKF->N, 0);
    for (int i = 0; i < pKF->N && i < (int)vpMatched.size(); i++) slot[i] = vpMatched[i] ? ORBM_MP_PRESENT : 0;   // :512
    vector<int32_t> owner;
    const int n = q.Run(2, fid, pKF->N, TH_LOW, false, false, false, slot.data(), owner);
    for (int idx = 0; idx < pKF->N; idx++)
        if (owner[idx] >= 0) vpMatched[idx] = vpPoints[owner[idx]];                       // :535
    return n;
}

// ---- ORBmatcher.cc:1468-1752 --------------------------------------------------------------------------------------
int ORBmatcher::SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint *> &vpMatches12, const float &s12, const cv::Mat &R12,
                             const cv::Mat &t12, const float th) {
    const float &fx = pKF1->fx, &fy = pKF1->fy, &cx = pKF1->cx, &cy = pKF1->cy;          // :1471-1474 (KF1's for both directions)
    cv::Mat R1w = pKF1->GetRotation(), t1w = pKF1->GetTranslation();
    cv::Mat R2w = pKF2->GetRotation(), t2w = pKF2->GetTranslation();
    // :1485-1487.  scalar * Mat is a float convertTo; -sR21*t12 is one gemm with alpha = -1
    float sR12[9], sR21[9], t21[3];
    const float *r12 = R12.ptr<float>();
    const float inv_s = (float)(1.0 / (double)s12);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { sR12[3 * r + c] = s12 * r12[3 * r + c]; sR21[3 * r + c] = r12[3 * c + r] * inv_s; }
    orbm_host::gemm3(sR21, 3, false, t12.ptr<float>(), -1.0, nullptr, t21);

    const vector<MapPoint *> vpMapPoints1 = pKF1->GetMapPointMatches();
    const int N1 = (int)vpMapPoints1.size();
    const vector<MapPoint *> vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N2 = (int)vpMapPoints2.size();
    vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
    for (int i = 0; i < N1; i++) {                                                        // :1500-1510
        MapPoint *pMP = vpMatches12[i];
        if (pMP) {
            vbAlreadyMatched1[i] = true;
            int idx2 = pMP->GetIndexInKeyFrame(pKF2);
            if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
        }
    }
    if (N1 == 0 || N2 == 0 || pKF1->N == 0 || pKF2->N == 0) return 0;

    // one direction: points of `src` (world -> its camera by Rsw, tsw -> the other camera by sR, t) searched in pKFdst
    auto build = [&](WinQ &q, const vector<MapPoint *> &pts, const vector<bool> &done, const cv::Mat &Rsw, const cv::Mat &tsw,
                     const float *sR, const float *t, KeyFrame *pKFdst) {
        for (size_t i = 0; i < pts.size(); i++) {
            MapPoint *pMP = pts[i];
            if (!pMP || done[i]) continue;                                                // :1527
            if (pMP->isBad()) continue;
            cv::Mat p3Dw = pMP->GetWorldPos();
            float pa[3], pb[3];
            orbm_host::gemm3(Rsw.ptr<float>(), 3, false, p3Dw.ptr<float>(), 1.0, tsw.ptr<float>(), pa);   // :1534
            orbm_host::gemm3(sR, 3, false, pa, 1.0, t, pb);                                                // :1535
            if (pb[2] < 0.0) continue;
            const float invz = 1.0 / pb[2];
            const float x = pb[0] * invz, y = pb[1] * invz;
            const float u = fx * x + cx, v = fy * y + cy;
            if (!pKFdst->IsInImage(u, v)) continue;
            const float maxDistance = pMP->GetMaxDistanceInvariance();
            const float minDistance = pMP->GetMinDistanceInvariance();
            const float dist3D = orbm_host::norm3(pb);                                    // :1562
            if (dist3D < minDistance || dist3D > maxDistance) continue;
            const int nPredictedLevel = pMP->PredictScale(dist3D, pKFdst);
            q.radius[i] = th * pKFdst->mvScaleFactors[nPredictedLevel];                   // :1582
            q.minL[i] = nPredictedLevel - 1; q.maxL[i] = nPredictedLevel;                 // :1608
            q.u[i] = u; q.v[i] = v;
            q.flags[i] = ORBM_Q_VALID | ORBM_Q_OBS;
            q.SetDesc((int)i, pMP->GetDescriptor());
        }
    };
    WinQ q1(N1), q2(N2);
    build(q1, vpMapPoints1, vbAlreadyMatched1, R1w, t1w, sR21, t21, pKF2);
    build(q2, vpMapPoints2, vbAlreadyMatched2, R2w, t2w, sR12, t12.ptr<float>(), pKF1);
    const int32_t id1 = Resident(pKF1, vpMapPoints1, false), id2 = Resident(pKF2, vpMapPoints2, false);
    vector<vector<int32_t>> best;
    RunFree({&q1, &q2}, {id2, id1}, TH_HIGH, false, best);                   // both directions, one launch
    const vector<int32_t> &vnMatch1 = best[0], &vnMatch2 = best[1];
    int nFound = 0;
    for (int i1 = 0; i1 < N1; i1++) {                                                     // :1736-1749
        int idx2 = vnMatch1[i1];
        if (idx2 >= 0) {
            int idx1 = vnMatch2[idx2];
            if (idx1 == i1) {
                vpMatches12[i1] = vpMapPoints2[idx2];
                nFound++;
            }
        }
    }
    return nFound;
}

namespace {
// the projection gates both Fuse overloads share (:1140-1183 / :1341-1386); ur is only read by the chi-square gate
void FuseQueries(WinQ &q, KeyFrame *pKF, const vector<MapPoint *> &pts, const vector<char> &skip, const float *Rcw, const float *tcw,
                 const float *Ow, float bf, float th) {
    const float &fx = pKF->fx, &fy = pKF->fy, &cx = pKF->cx, &cy = pKF->cy;
    for (size_t i = 0; i < pts.size(); i++) {
        MapPoint *pMP = pts[i];
        if (skip[i]) continue;
        cv::Mat p3Dw = pMP->GetWorldPos();
        float p3Dc[3];
        orbm_host::gemm3(Rcw, 3, false, p3Dw.ptr<float>(), 1.0, tcw, p3Dc);
        if (p3Dc[2] < 0.0f) continue;
        const float invz = 1 / p3Dc[2];
        const float x = p3Dc[0] * invz, y = p3Dc[1] * invz;
        const float u = fx * x + cx, v = fy * y + cy;
        if (!pKF->IsInImage(u, v)) continue;
        const float ur = u - bf * invz;
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        const float *pw = p3Dw.ptr<float>();
        const float PO[3] = {pw[0] - Ow[0], pw[1] - Ow[1], pw[2] - Ow[2]};
        const float dist3D = orbm_host::norm3(PO);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        cv::Mat Pn = pMP->GetNormal();
        if (orbm_host::dot3(PO, Pn.ptr<float>()) < 0.5 * dist3D) continue;
        const int nPredictedLevel = pMP->PredictScale(dist3D, pKF);
        q.radius[i] = th * pKF->mvScaleFactors[nPredictedLevel];
        q.minL[i] = nPredictedLevel - 1; q.maxL[i] = nPredictedLevel;
        q.u[i] = u; q.v[i] = v; q.ur[i] = ur;
        q.flags[i] = ORBM_Q_VALID | ORBM_Q_OBS;
        q.SetDesc((int)i, pMP->GetDescriptor());
    }
}
}  // namespace

// ---- ORBmatcher.cc:1111-1288 --------------------------------------------------------------------------------------
int ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint *> &vpMapPoints, const float th) {
    const int nMPs = (int)vpMapPoints.size();
    if (nMPs == 0 || pKF->N == 0) return 0;
    cv::Mat Rcw = pKF->GetRotation(), tcw = pKF->GetTranslation(), Ow = pKF->GetCameraCenter();
    // A point that fails `!pMP || isBad() || IsInKeyFrame(pKF)` (:1134-1138) now also fails it when the reference's loop
    // reaches it: Replace only ever makes points bad or adds them to pKF.  So it is safe to drop them before the launch.
    vector<char> skip(nMPs, 0);
    for (int i = 0; i < nMPs; i++) {
        MapPoint *pMP = vpMapPoints[i];
        skip[i] = !pMP || pMP->isBad() || pMP->IsInKeyFrame(pKF);
    }
    WinQ q(nMPs);
    FuseQueries(q, pKF, vpMapPoints, skip, Rcw.ptr<float>(), tcw.ptr<float>(), Ow.ptr<float>(), pKF->mbf, th);
    const int32_t fid = Resident(pKF);
    vector<vector<int32_t>> best;
    RunFree({&q}, {fid}, TH_LOW, true, best);
    // replay of the reference loop, in order, with the state-dependent gates evaluated now
    int nFused = 0;
    for (int i = 0; i < nMPs; i++) {
        MapPoint *pMP = vpMapPoints[i];
        if (!pMP) continue;
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;                             // :1137
        const int bestIdx = best[0][i];
        if (bestIdx < 0) continue;                                                        // gates, empty window or bestDist > TH_LOW
        MapPoint *pMPinKF = pKF->GetMapPoint(bestIdx);                                    // :1259
        if (pMPinKF) {
            if (!pMPinKF->isBad()) {
                if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                else pMPinKF->Replace(pMP);
            }
        } else {
            pMP->AddObservation(pKF, bestIdx);
            pKF->AddMapPoint(pMP, bestIdx);
        }
        nFused++;
    }
    return nFused;
}

// ---- ORBmatcher.cc:1301-1443 --------------------------------------------------------------------------------------
int ORBmatcher::Fuse(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint *> &vpPoints, float th, vector<MapPoint *> &vpReplacePoint) {
    const int nPoints = (int)vpPoints.size();
    if (nPoints == 0 || pKF->N == 0) return 0;
    float Rcw[9], tcw[3], Ow[3];
    DecomposeSim3(Scw, Rcw, tcw, Ow);                                                     // :1313-1317
    const set<MapPoint *> spAlreadyFound = pKF->GetMapPoints();                           // :1320, a snapshot in the reference too
    vector<char> skip(nPoints, 0);
    for (int i = 0; i < nPoints; i++) skip[i] = vpPoints[i]->isBad() || spAlreadyFound.count(vpPoints[i]);   // :1337 (nothing here makes a point bad)
    WinQ q(nPoints);
    FuseQueries(q, pKF, vpPoints, skip, Rcw, tcw, Ow, 0.f, th);
    const int32_t fid = Resident(pKF);
    vector<vector<int32_t>> best;
    RunFree({&q}, {fid}, TH_LOW, false, best);
    int nFused = 0;
    for (int iMP = 0; iMP < nPoints; iMP++) {
        const int bestIdx = best[0][iMP];
        if (bestIdx < 0) continue;
        MapPoint *pMP = vpPoints[iMP];
        MapPoint *pMPinKF = pKF->GetMapPoint(bestIdx);                                    // :1423, sees slots filled earlier in this loop
        if (pMPinKF) {
            if (!pMPinKF->isBad()) vpReplacePoint[iMP] = pMPinKF;
        } else {
            pMP->AddObservation(pKF, bestIdx);
            pKF->AddMapPoint(pMP, bestIdx);
        }
        nFused++;
    }
    return nFused;
}

// ---- MapPoint.cc:363-449, batched -------------------------------------------------------------------------------------
void ComputeDistinctiveDescriptorsBatch(const vector<MapPoint *> &vpMapPoints) {
    const int P = (int)vpMapPoints.size();
    if (P == 0) return;
    vector<int32_t> off(P + 1, 0);
    vector<uint8_t> desc;
    vector<const uint8_t *> rows;                 // source row of every gathered descriptor
    for (int p = 0; p < P; p++) {
        MapPoint *pMP = vpMapPoints[p];
        if (pMP && !pMP->isBad())                                                         // :375-376
            for (auto mit = pMP->mObservations.begin(), mend = pMP->mObservations.end(); mit != mend; mit++) {
                KeyFrame *pKF = mit->first;
                if (!pKF->isBad()) rows.push_back(pKF->mDescriptors.ptr<uint8_t>((int)mit->second));   // :397-398
            }
        off[p + 1] = (int32_t)rows.size();
    }
    desc.resize(rows.size() * 32);
    for (size_t i = 0; i < rows.size(); i++) memcpy(&desc[i * 32], rows[i], 32);
    vector<int32_t> best(P, -1);
    Check(orbm_distinctive_descriptors_batch(Ctx(), P, off.data(), desc.empty() ? nullptr : desc.data(), best.data(), 0),
          "orbm_distinctive_descriptors_batch");
    for (int p = 0; p < P; p++)
        if (best[p] >= 0) {                                                               // empty list: mDescriptor untouched (:405-406)
            cv::Mat d(1, 32, CV_8U);
            memcpy(d.ptr<uint8_t>(), rows[off[p] + best[p]], 32);
            vpMapPoints[p]->mDescriptor = d;                                              // :447
        }
}

}  // namespace ORB_SLAM2
