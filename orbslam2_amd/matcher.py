"""Python host layer over the C ABI (include/orbmatch.h): one `Context` per calling thread / GPU.

Methods mirror the ORB_SLAM2::ORBmatcher searches in their flat, batched form.  The numpy forms
stage index lists and outputs through the library (io_on_device = 0); the `*_dev` forms take raw
device addresses (e.g. torch tensors' data_ptr()) and only enqueue work on the context's stream.
"""
import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib
from .views import CBowVecView, CFrameView, CProjQueries, CWinQueries, FrameData, ProjQueries, WinQueries, GRID_CELLS


class OrbmError(RuntimeError):
    pass


def _addr(a) -> Optional[int]:
    """host numpy array -> address (keeps no reference; caller holds the array)"""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


class Context:
    def __init__(self, device: int = 0):
        self._L = _lib.load()
        self._h = self._L.orbm_create(device)
        if not self._h:
            raise OrbmError("orbm_create failed: " + self._L.orbm_last_error(None).decode())
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._L.orbm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int) -> int:
        if rc < 0:
            raise OrbmError(f"orbm error {rc}: " + self._L.orbm_last_error(self._h).decode())
        return rc

    # ---- context ----
    def set_stream(self, hip_stream: Optional[int]):
        """Adopt a caller stream (0 / None = the HIP null stream, e.g. torch's default stream)."""
        self._chk(self._L.orbm_set_stream(self._h, hip_stream or None))

    def use_own_stream(self):
        self._chk(self._L.orbm_set_stream(self._h, C.c_void_p(-1)))

    def synchronize(self):
        self._chk(self._L.orbm_synchronize(self._h))

    def device_info(self):
        ncu, clk = C.c_int32(0), C.c_int32(0)
        name = C.create_string_buffer(256)
        self._chk(self._L.orbm_device_info(self._h, C.byref(ncu), C.byref(clk), name, 256))
        return {"n_cu": ncu.value, "clock_khz": clk.value, "name": name.value.decode()}

    def popcount_peak(self, iters: int = 4096) -> float:
        r = C.c_double(0)
        self._chk(self._L.orbm_popcount_peak(self._h, iters, C.byref(r)))
        return r.value

    # ---- storage ----
    def upload_frames(self, frames: Sequence[FrameData]) -> np.ndarray:
        n = len(frames)
        arr = (CFrameView * max(n, 1))()
        keep = []
        for i, f in enumerate(frames):
            v = f.view()
            keep.append(v)
            arr[i] = v
        ids = np.zeros(max(n, 1), np.int32)
        self._chk(self._L.orbm_frames_upload(self._h, n, arr, _addr(ids)))
        return ids[:n]

    def free_frames(self, ids):
        ids = np.ascontiguousarray(ids, np.int32)
        self._chk(self._L.orbm_frames_free(self._h, len(ids), _addr(ids)))

    def set_mp_flags(self, fid: int, flags: np.ndarray):
        flags = np.ascontiguousarray(flags, np.uint8)
        self._chk(self._L.orbm_frame_set_mp_flags(self._h, int(fid), _addr(flags)))

    def grid(self, fid: int, n: int):
        off = np.zeros(GRID_CELLS + 1, np.uint32)
        idx = np.zeros(max(n, 1), np.uint32)
        self._chk(self._L.orbm_frame_grid_download(self._h, int(fid), _addr(off), _addr(idx)))
        return off, idx[: int(off[-1])]

    def upload_proj_queries(self, sets: Sequence[ProjQueries]) -> np.ndarray:
        n = len(sets)
        arr = (CProjQueries * max(n, 1))()
        keep = [s.view() for s in sets]
        for i, v in enumerate(keep):
            arr[i] = v
        ids = np.zeros(max(n, 1), np.int32)
        self._chk(self._L.orbm_proj_queries_upload(self._h, n, arr, _addr(ids)))
        return ids[:n]

    def upload_win_queries(self, sets: Sequence[WinQueries]) -> np.ndarray:
        n = len(sets)
        arr = (CWinQueries * max(n, 1))()
        keep = [s.view() for s in sets]
        for i, v in enumerate(keep):
            arr[i] = v
        ids = np.zeros(max(n, 1), np.int32)
        self._chk(self._L.orbm_win_queries_upload(self._h, n, arr, _addr(ids)))
        return ids[:n]

    def free_queries(self, ids):
        ids = np.ascontiguousarray(ids, np.int32)
        self._chk(self._L.orbm_queries_free(self._h, len(ids), _addr(ids)))

    # ---- DBoW2 transform (Frame::ComputeBoW / KeyFrame::ComputeBoW), FeatureVector kept on the device ----
    def upload_vocabulary(self, vocab) -> int:
        v = vocab.view()
        vid = C.c_int32(-1)
        self._chk(self._L.orbm_vocab_upload(self._h, C.byref(v), C.byref(vid)))
        return int(vid.value)

    def free_vocabulary(self, vid: int):
        self._chk(self._L.orbm_vocab_free(self._h, int(vid)))

    def compute_bow(self, vid: int, frames, levelsup: int = 4, stride: int = 0, want_ids: bool = True, want_bowvecs: bool = False):
        """The frames' FeatureVectors are replaced on the device.  -> (word_id, node_id) [n_frames, stride] when want_ids;
        with want_bowvecs also (or only) the ids of the device-built BowVectors."""
        fr = np.ascontiguousarray(frames, np.int32)
        p = len(fr)
        bv = np.full(max(p, 1), -1, np.int32) if want_bowvecs else None
        word = np.full((max(p, 1), max(stride, 1)), -1, np.int32) if want_ids else None
        node = np.full((max(p, 1), max(stride, 1)), -1, np.int32) if want_ids else None
        self._chk(self._L.orbm_frames_compute_bow(self._h, int(vid), p, _addr(fr), int(levelsup), int(stride), _addr(word), _addr(node),
                                                  _addr(bv)))
        out = ()
        if want_ids:
            out += (word[:p], node[:p])
        if want_bowvecs:
            out += (bv[:p],)
        return out if out else None

    def download_bowvec(self, bid: int, capacity: int = 16384):
        n = C.c_int32(0)
        word = np.zeros(capacity, np.uint32); value = np.zeros(capacity, np.float64)
        self._chk(self._L.orbm_bowvec_download(self._h, int(bid), C.byref(n), _addr(word), _addr(value), capacity))
        return word[: n.value].copy(), value[: n.value].copy()

    def transform(self, vid: int, desc: np.ndarray, levelsup: int = 4):
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(d)
        word = np.full(max(n, 1), -1, np.int32)
        node = np.full(max(n, 1), -1, np.int32)
        self._chk(self._L.orbm_vocab_transform(self._h, int(vid), n, _addr(d) if n else None, int(levelsup), _addr(word), _addr(node), 0))
        return word[:n], node[:n]

    def feature_vector(self, fid: int, n: int):
        fv_node = np.zeros(max(n, 1), np.uint32)
        fv_off = np.zeros(n + 1, np.uint32)
        fv_idx = np.zeros(max(n, 1), np.uint32)
        nn = C.c_int32(0)
        self._chk(self._L.orbm_frame_fv_download(self._h, int(fid), C.byref(nn), _addr(fv_node), _addr(fv_off), _addr(fv_idx)))
        k = int(nn.value)
        return fv_node[:k], fv_off[:k + 1], fv_idx[: int(fv_off[k])]

    # ---- KeyFrameDatabase scoring (DBoW2 L1 score between BowVectors) ----
    def upload_bowvecs(self, vecs) -> np.ndarray:
        n = len(vecs)
        arr = (CBowVecView * max(n, 1))()
        keep = [v.view() for v in vecs]
        for i, v in enumerate(keep):
            arr[i] = v
        ids = np.zeros(max(n, 1), np.int32)
        self._chk(self._L.orbm_bowvecs_upload(self._h, n, arr, _addr(ids)))
        return ids[:n]

    def free_bowvecs(self, ids):
        ids = np.ascontiguousarray(ids, np.int32)
        self._chk(self._L.orbm_bowvecs_free(self._h, len(ids), _addr(ids)))

    def bow_score(self, va, vb):
        va = np.ascontiguousarray(va, np.int32); vb = np.ascontiguousarray(vb, np.int32)
        p = len(va)
        score = np.zeros(max(p, 1), np.float64)
        common = np.zeros(max(p, 1), np.int32)
        self._chk(self._L.orbm_bow_score_batch(self._h, p, _addr(va), _addr(vb), _addr(score), _addr(common), 0))
        return score[:p], common[:p]

    def bow_score_dev(self, n_pairs, va_ptr, vb_ptr, score_ptr, common_ptr):
        self._chk(self._L.orbm_bow_score_batch(self._h, n_pairs, va_ptr, vb_ptr, score_ptr, common_ptr, 1))

    # ---- a1 ----
    def hamming_pairs(self, a: np.ndarray, b: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(a, np.uint8).reshape(-1, 32)
        b = np.ascontiguousarray(b, np.uint8).reshape(-1, 32)
        assert a.shape == b.shape
        out = np.zeros(max(a.shape[0], 1), np.int32)
        self._chk(self._L.orbm_hamming256_pairs(self._h, a.shape[0], _addr(a), _addr(b), _addr(out), 0))
        return out[: a.shape[0]]

    # ---- MapPoint::ComputeDistinctiveDescriptors, batched ----
    def distinctive_descriptors(self, off: np.ndarray, desc: np.ndarray) -> np.ndarray:
        off = np.ascontiguousarray(off, np.int32)
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(off) - 1
        out = np.full(max(n, 1), -1, np.int32)
        self._chk(self._L.orbm_distinctive_descriptors_batch(self._h, n, _addr(off), _addr(desc) if len(desc) else None,
                                                             _addr(out), 0))
        return out[:n]

    # ---- descriptor stage of Frame::ComputeStereoMatches, batched ----
    def stereo_match(self, left, right, max_disparity, stride: int):
        fl = np.ascontiguousarray(left, np.int32)
        fr = np.ascontiguousarray(right, np.int32)
        md = np.ascontiguousarray(max_disparity, np.float32)
        p = len(fl)
        idx = np.full((max(p, 1), max(stride, 1)), -1, np.int32)
        dist = np.full((max(p, 1), max(stride, 1)), 100, np.int32)
        self._chk(self._L.orbm_stereo_match_batch(self._h, p, _addr(fl), _addr(fr), _addr(md), stride, _addr(idx), _addr(dist), 0))
        return idx[:p], dist[:p]

    # ---- config 2 ----
    def bf2nn(self, fa, fb, ratio: float, th: int, stride: int, want_dist: bool = True):
        fa = np.ascontiguousarray(fa, np.int32)
        fb = np.ascontiguousarray(fb, np.int32)
        p = len(fa)
        match = np.full((max(p, 1), max(stride, 1)), -1, np.int32)
        bd = np.full_like(match, 256) if want_dist else None
        sd = np.full_like(match, 256) if want_dist else None
        counts = np.zeros(max(p, 1), np.int32)
        self._chk(self._L.orbm_bf2nn_batch(self._h, p, _addr(fa), _addr(fb), ratio, th, stride, _addr(match),
                                           _addr(bd), _addr(sd), _addr(counts), 0))
        return counts[:p], match[:p], (bd[:p] if want_dist else None), (sd[:p] if want_dist else None)

    def bf2nn_dev(self, n_pairs, fa_ptr, fb_ptr, ratio, th, stride, match_ptr, best_ptr, second_ptr, counts_ptr):
        self._chk(self._L.orbm_bf2nn_batch(self._h, n_pairs, fa_ptr, fb_ptr, ratio, th, stride, match_ptr, best_ptr,
                                           second_ptr, counts_ptr, 1))

    def bf2nn_prepare(self):
        """Builds the +-1 images of every live frame now, so that later device-path bf2nn calls only enqueue work."""
        self._chk(self._L.orbm_bf2nn_prepare(self._h))

    # ---- config 5: match + compact records ----
    SUMMARY_DTYPE = np.dtype([("pair_id", np.uint32), ("n", np.uint16), ("pad", np.uint16)])

    def pack_records(self, pair_id, counts, match, threshold: int, list_cap: int):
        counts = np.ascontiguousarray(counts, np.int32); match = np.ascontiguousarray(match, np.int32)
        p, stride = match.shape
        pid = None if pair_id is None else np.ascontiguousarray(pair_id, np.uint32)
        summ = np.zeros(max(p, 1), self.SUMMARY_DTYPE); lists = np.zeros(max(list_cap, 1), np.uint32); totals = np.zeros(2, np.uint32)
        self._chk(self._L.orbm_pack_records(self._h, p, _addr(pid), _addr(counts), _addr(match), stride, threshold, _addr(summ),
                                            _addr(lists) if list_cap else None, list_cap, _addr(totals), 0))
        return summ[:p], lists[: min(int(totals[1]), list_cap)], totals

    def allpairs_batch(self, kind: int, fa, fb, pair_id, ratio: float, th: int, check_ori: bool, stride: int, threshold: int,
                       list_cap: int):
        fa = np.ascontiguousarray(fa, np.int32); fb = np.ascontiguousarray(fb, np.int32)
        p = len(fa)
        pid = None if pair_id is None else np.ascontiguousarray(pair_id, np.uint32)
        summ = np.zeros(max(p, 1), self.SUMMARY_DTYPE); lists = np.zeros(max(list_cap, 1), np.uint32); totals = np.zeros(2, np.uint32)
        self._chk(self._L.orbm_allpairs_batch(self._h, kind, p, _addr(fa), _addr(fb), _addr(pid), ratio, th, int(check_ori), stride,
                                              threshold, _addr(summ), _addr(lists) if list_cap else None, list_cap, _addr(totals), 0))
        return summ[:p], lists[: min(int(totals[1]), list_cap)], totals

    @staticmethod
    def allpairs_multi(ctxs, handles, kind: int, kf_a, kf_b, pair_id, ratio: float, th: int, check_ori: bool, stride: int,
                       threshold: int, list_cap: int):
        """orbm_allpairs: one pair list over several contexts (GPUs) from this process; handles[c][k] = handle of keyframe k
        in ctxs[c].  Returns what allpairs_batch returns for the whole list."""
        import ctypes as C
        L = ctxs[0]._L
        n_ctx = len(ctxs)
        carr = (C.c_void_p * n_ctx)(*[c._h for c in ctxs])
        hs = [np.ascontiguousarray(h, np.int32) for h in handles]
        harr = (C.c_void_p * n_ctx)(*[h.ctypes.data for h in hs])
        kf_a = np.ascontiguousarray(kf_a, np.int32); kf_b = np.ascontiguousarray(kf_b, np.int32)
        p = len(kf_a)
        pid = None if pair_id is None else np.ascontiguousarray(pair_id, np.uint32)
        summ = np.zeros(max(p, 1), Context.SUMMARY_DTYPE); lists = np.zeros(max(list_cap, 1), np.uint32); totals = np.zeros(2, np.uint32)
        rc = L.orbm_allpairs(carr, n_ctx, harr, kind, p, _addr(kf_a), _addr(kf_b), _addr(pid), ratio, th, int(check_ori), stride,
                             threshold, _addr(summ), _addr(lists) if list_cap else None, list_cap, _addr(totals))
        ctxs[0]._chk(rc)
        return summ[:p], lists[: min(int(totals[1]), list_cap)], totals

    def allpairs_batch_dev(self, kind, n_pairs, fa_ptr, fb_ptr, pair_id_ptr, ratio, th, check_ori, stride, threshold,
                           summaries_ptr, lists_ptr, list_cap, totals_ptr):
        self._chk(self._L.orbm_allpairs_batch(self._h, kind, n_pairs, fa_ptr, fb_ptr, pair_id_ptr, ratio, th, int(check_ori), stride,
                                              threshold, summaries_ptr, lists_ptr, list_cap, totals_ptr, 1))

    # ---- a7 / a8 ----
    def search_bow(self, variant: int, fa, fb, ratio: float, check_ori: bool, stride: int):
        fa = np.ascontiguousarray(fa, np.int32)
        fb = np.ascontiguousarray(fb, np.int32)
        p = len(fa)
        match = np.full((max(p, 1), max(stride, 1)), -1, np.int32)
        counts = np.zeros(max(p, 1), np.int32)
        self._chk(self._L.orbm_search_bow_batch(self._h, variant, p, _addr(fa), _addr(fb), ratio, int(check_ori),
                                                stride, _addr(match), _addr(counts), 0))
        return counts[:p], match[:p]

    def search_bow_dev(self, variant, n_pairs, fa_ptr, fb_ptr, ratio, check_ori, stride, match_ptr, counts_ptr):
        self._chk(self._L.orbm_search_bow_batch(self._h, variant, n_pairs, fa_ptr, fb_ptr, ratio, int(check_ori),
                                                stride, match_ptr, counts_ptr, 1))

    def search_bow_noalloc(self, variant, n_pairs, fa_ptr, fb_ptr, ratio, check_ori, stride, match_ptr, counts_ptr):
        """Allocation-free variant (device pointers only, at most 2048 pairs; static scratch buffer inside the library)."""
        self._chk(self._L.orbm_search_bow_batch_noalloc(self._h, variant, n_pairs, fa_ptr, fb_ptr, ratio, int(check_ori),
                                                        stride, match_ptr, counts_ptr))

    # ---- a9 ----
    def search_triang(self, fa, fb, f12, epipole, only_stereo: bool, check_ori: bool, stride: int):
        fa = np.ascontiguousarray(fa, np.int32)
        fb = np.ascontiguousarray(fb, np.int32)
        p = len(fa)
        f12 = np.ascontiguousarray(f12, np.float32).reshape(p, 9)
        ep = np.ascontiguousarray(epipole, np.float32).reshape(p, 2)
        match = np.full((max(p, 1), max(stride, 1)), -1, np.int32)
        counts = np.zeros(max(p, 1), np.int32)
        self._chk(self._L.orbm_search_triang_batch(self._h, p, _addr(fa), _addr(fb), _addr(f12), _addr(ep),
                                                   int(only_stereo), int(check_ori), stride, _addr(match),
                                                   _addr(counts), 0))
        return counts[:p], match[:p]

    # ---- a6 ----
    def search_proj_points(self, frames, qsets, th: float, ratio: float, stride: int, slot_flags=None):
        fr = np.ascontiguousarray(frames, np.int32)
        qs = np.ascontiguousarray(qsets, np.int32)
        k = len(fr)
        owner = np.full((max(k, 1), max(stride, 1)), -1, np.int32)
        counts = np.zeros(max(k, 1), np.int32)
        sf = None
        if slot_flags is not None:
            sf = np.ascontiguousarray(slot_flags, np.uint8)
            assert sf.shape == (k, stride)
        self._chk(self._L.orbm_search_proj_points_batch(self._h, k, _addr(fr), _addr(qs), th, ratio, _addr(sf),
                                                        stride, _addr(owner), _addr(counts), 0))
        return counts[:k], owner[:k]

    def search_proj_points_once(self, frame: int, queries: ProjQueries, th: float, ratio: float, n: int, slot_flags=None):
        """One search, the query set travelling with the call (orbm_search_proj_points_once): -> (count, owner[n])"""
        import ctypes as C
        v = queries.view()
        owner = np.full(max(n, 1), -1, np.int32); cnt = np.zeros(1, np.int32)
        sf = None if slot_flags is None else np.ascontiguousarray(slot_flags, np.uint8)
        self._chk(self._L.orbm_search_proj_points_once(self._h, int(frame), C.byref(v), th, ratio, _addr(sf), _addr(owner), _addr(cnt)))
        return int(cnt[0]), owner[:n]

    def search_window_once(self, mode: int, frame: int, queries: WinQueries, dist_th: int, check_ori: bool, n: int, slot_flags=None):
        """orbm_search_window_once: -> (count, owner[n])"""
        import ctypes as C
        v = queries.view()
        owner = np.full(max(n, 1), -1, np.int32); cnt = np.zeros(1, np.int32)
        sf = None if slot_flags is None else np.ascontiguousarray(slot_flags, np.uint8)
        self._chk(self._L.orbm_search_window_once(self._h, mode, int(frame), C.byref(v), dist_th, int(check_ori), _addr(sf), _addr(owner), _addr(cnt)))
        return int(cnt[0]), owner[:n]

    def search_proj_points_dev(self, n_inst, frame_ptr, qset_ptr, th, ratio, slot_flags_ptr, stride, owner_ptr,
                               counts_ptr):
        self._chk(self._L.orbm_search_proj_points_batch(self._h, n_inst, frame_ptr, qset_ptr, th, ratio,
                                                        slot_flags_ptr, stride, owner_ptr, counts_ptr, 1))

    # ---- a11 ----
    def search_window(self, mode: int, frames, qsets, dist_th: int, check_ori: bool, stride: int, slot_flags=None):
        fr = np.ascontiguousarray(frames, np.int32)
        qs = np.ascontiguousarray(qsets, np.int32)
        k = len(fr)
        owner = np.full((max(k, 1), max(stride, 1)), -1, np.int32)
        counts = np.zeros(max(k, 1), np.int32)
        sf = None
        if slot_flags is not None:
            sf = np.ascontiguousarray(slot_flags, np.uint8)
            assert sf.shape == (k, stride)
        self._chk(self._L.orbm_search_window_batch(self._h, mode, k, _addr(fr), _addr(qs), dist_th, int(check_ori),
                                                   _addr(sf), stride, _addr(owner), _addr(counts), 0))
        return counts[:k], owner[:k]

    # ---- a12 ----
    def search_window_free(self, frames, qsets, dist_th: int, chi2_gate: bool, stride_q: int):
        fr = np.ascontiguousarray(frames, np.int32)
        qs = np.ascontiguousarray(qsets, np.int32)
        k = len(fr)
        best = np.full((max(k, 1), max(stride_q, 1)), -1, np.int32)
        self._chk(self._L.orbm_search_window_free_batch(self._h, k, _addr(fr), _addr(qs), dist_th, int(chi2_gate),
                                                        stride_q, _addr(best), 0))
        return best[:k]

    def search_window_free_once(self, frames, query_sets, dist_th: int, chi2_gate: bool, stride_q: int):
        """orbm_search_window_free_once: the query sets travel with the call -> best[k, stride_q]"""
        from .views import CWinQueries
        fr = np.ascontiguousarray(frames, np.int32)
        k = len(fr)
        views = (CWinQueries * max(k, 1))(*[q.view() for q in query_sets])
        best = np.full((max(k, 1), max(stride_q, 1)), -1, np.int32)
        self._chk(self._L.orbm_search_window_free_once(self._h, k, _addr(fr), views, dist_th, int(chi2_gate), stride_q, _addr(best)))
        return best[:k]

    # ---- a5 ----
    def search_init(self, f1: int, f2: int, prev_xy: np.ndarray, window: int, ratio: float, check_ori: bool):
        prev = np.ascontiguousarray(prev_xy, np.float32).copy()
        n1 = prev.shape[0]
        m12 = np.full(max(n1, 1), -1, np.int32)
        cnt = self._chk(self._L.orbm_search_init(self._h, int(f1), int(f2), _addr(prev), window, ratio,
                                                 int(check_ori), _addr(m12)))
        return cnt, m12[:n1], prev
