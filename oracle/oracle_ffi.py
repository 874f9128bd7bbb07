"""ctypes binding of oracle/liborb_oracle.so (the CPU restatement; TEST INFRASTRUCTURE ONLY).

Importable only from tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke(); nothing under
orbslam2_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from orbslam2_amd.views import CBowVecView, CFrameView, CProjQueries, CVocabView, CWinQueries, GRID_CELLS

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liborb_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "orb_oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def timing_build():
    """CPU-baseline build (bench.py only): the same source with -O3 -march=native -ffp-contract=off, compiled on the
    machine that runs it (the committed -O2 -msse2 library stays the parity checker) and made the library in use.
    Returns the flags, or None if no compiler is there (the -O2 build stays in use)."""
    import tempfile
    global _lib
    flags = ["-O3", "-march=native", "-std=c++17", "-fPIC", "-ffp-contract=off"]
    out = os.path.join(tempfile.gettempdir(), f"liborb_oracle_native_{os.getuid()}_{os.getpid()}.so")
    try:
        subprocess.check_call(["g++", *flags, "-shared", "-o", out, os.path.join(_HERE, "orb_oracle.cpp")], timeout=300,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _lib = _bind(C.CDLL(out))
        os.unlink(out)
        return " ".join(flags)
    except Exception:  # noqa: BLE001
        return None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = _bind(C.CDLL(_SO))
    return _lib


def _bind(L):
    u8p, i32p, u32p, f32p = (C.POINTER(t) for t in (C.c_uint8, C.c_int32, C.c_uint32, C.c_float))
    u64p = C.POINTER(C.c_uint64)
    FV, PQ, WQ = C.POINTER(CFrameView), C.POINTER(CProjQueries), C.POINTER(CWinQueries)
    L.orc_set_popcnt.argtypes = [C.c_int]
    L.orc_hamming256.argtypes = [u8p, u8p]
    L.orc_hamming256_popcnt.argtypes = [u8p, u8p]
    L.orc_three_maxima.argtypes = [i32p, C.c_int, i32p]
    L.orc_rot_bin.argtypes = [C.c_float, C.c_float]
    L.orc_build_grid.argtypes = [FV, u32p, u32p]
    L.orc_features_in_area.argtypes = [FV, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, i32p]
    L.orc_bf2nn.argtypes = [FV, FV, C.c_float, C.c_int, i32p, i32p, i32p, u64p]
    L.orc_search_bow.argtypes = [C.c_int, FV, FV, C.c_float, C.c_int, i32p, u64p]
    L.orc_search_triang.argtypes = [FV, FV, f32p, f32p, C.c_int, C.c_int, i32p, u64p]
    L.orc_search_proj_points.argtypes = [FV, PQ, C.c_float, C.c_float, u8p, i32p, u64p]
    L.orc_search_window.argtypes = [C.c_int, FV, WQ, C.c_int, C.c_int, u8p, i32p, u64p]
    L.orc_search_window_free.argtypes = [FV, WQ, C.c_int, C.c_int, i32p, u64p]
    L.orc_search_init.argtypes = [FV, FV, f32p, C.c_int, C.c_float, C.c_int, i32p, u64p]
    L.orc_stereo_match.argtypes = [FV, FV, C.c_int, C.c_float, i32p, i32p, u64p]
    L.orc_distinctive_descriptor.argtypes = [u8p, C.c_int]
    L.orc_transform.argtypes = [C.POINTER(CVocabView), u8p, C.c_int, C.c_int, i32p, i32p, u32p, u32p, u32p, i32p, u32p, C.POINTER(C.c_double)]
    L.orc_bow_score.argtypes = [C.POINTER(CBowVecView), C.POINTER(CBowVecView), i32p]
    L.orc_bow_score.restype = C.c_double
    L.orc_predict_scale.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int]
    return L


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def set_popcnt(on: bool):
    lib().orc_set_popcnt(int(on))


def hamming256(a: np.ndarray, b: np.ndarray, popcnt=False) -> int:
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    f = lib().orc_hamming256_popcnt if popcnt else lib().orc_hamming256
    return int(f(_p(a, C.c_uint8), _p(b, C.c_uint8)))


def three_maxima(sizes) -> tuple:
    s = np.ascontiguousarray(sizes, np.int32)
    out = np.zeros(3, np.int32)
    lib().orc_three_maxima(_p(s, C.c_int32), len(s), _p(out, C.c_int32))
    return tuple(int(v) for v in out)


def rot_bin(a1: float, a2: float) -> int:
    return int(lib().orc_rot_bin(a1, a2))


def build_grid(f):
    off = np.zeros(GRID_CELLS + 1, np.uint32)
    idx = np.zeros(max(f.n, 1), np.uint32)
    v = f.view()
    n = lib().orc_build_grid(C.byref(v), _p(off, C.c_uint32), _p(idx, C.c_uint32))
    return off, idx[:n]


def features_in_area(f, x, y, r, min_level=-1, max_level=-1, kf_mode=False):
    out = np.zeros(max(f.n, 1), np.int32)
    v = f.view()
    n = lib().orc_features_in_area(C.byref(v), x, y, r, min_level, max_level, int(kf_mode), _p(out, C.c_int32))
    return out[:n].copy()


def bf2nn(a, b, ratio, th):
    match = np.zeros(max(a.n, 1), np.int32)
    bd = np.zeros(max(a.n, 1), np.int32)
    sd = np.zeros(max(a.n, 1), np.int32)
    ncmp = C.c_uint64(0)
    va, vb = a.view(), b.view()
    cnt = lib().orc_bf2nn(C.byref(va), C.byref(vb), ratio, th, _p(match, C.c_int32), _p(bd, C.c_int32),
                          _p(sd, C.c_int32), C.byref(ncmp))
    return int(cnt), match[:a.n], bd[:a.n], sd[:a.n], int(ncmp.value)


def search_bow(variant, a, b, ratio, check_ori):
    n_out = b.n if variant == 0 else a.n
    match = np.zeros(max(n_out, 1), np.int32)
    ncmp = C.c_uint64(0)
    va, vb = a.view(), b.view()
    cnt = lib().orc_search_bow(variant, C.byref(va), C.byref(vb), ratio, int(check_ori), _p(match, C.c_int32),
                               C.byref(ncmp))
    return int(cnt), match[:n_out], int(ncmp.value)


def search_triang(a, b, f12, epipole, only_stereo, check_ori):
    match = np.zeros(max(a.n, 1), np.int32)
    ncmp = C.c_uint64(0)
    f12 = np.ascontiguousarray(f12, np.float32)
    ep = np.ascontiguousarray(epipole, np.float32)
    va, vb = a.view(), b.view()
    cnt = lib().orc_search_triang(C.byref(va), C.byref(vb), _p(f12, C.c_float), _p(ep, C.c_float),
                                  int(only_stereo), int(check_ori), _p(match, C.c_int32), C.byref(ncmp))
    return int(cnt), match[:a.n], int(ncmp.value)


def search_proj_points(f, q, th, ratio, slot_flags=None):
    owner = np.zeros(max(f.n, 1), np.int32)
    ncmp = C.c_uint64(0)
    sf = None if slot_flags is None else np.ascontiguousarray(slot_flags, np.uint8)
    vf, vq = f.view(), q.view()
    cnt = lib().orc_search_proj_points(C.byref(vf), C.byref(vq), th, ratio, _p(sf, C.c_uint8),
                                       _p(owner, C.c_int32), C.byref(ncmp))
    return int(cnt), owner[:f.n], int(ncmp.value)


def search_window(mode, f, q, dist_th, check_ori, slot_flags=None):
    owner = np.zeros(max(f.n, 1), np.int32)
    ncmp = C.c_uint64(0)
    sf = None if slot_flags is None else np.ascontiguousarray(slot_flags, np.uint8)
    vf, vq = f.view(), q.view()
    cnt = lib().orc_search_window(mode, C.byref(vf), C.byref(vq), dist_th, int(check_ori), _p(sf, C.c_uint8),
                                  _p(owner, C.c_int32), C.byref(ncmp))
    return int(cnt), owner[:f.n], int(ncmp.value)


def search_window_free(f, q, dist_th, chi2_gate):
    best = np.zeros(max(q.n, 1), np.int32)
    ncmp = C.c_uint64(0)
    vf, vq = f.view(), q.view()
    lib().orc_search_window_free(C.byref(vf), C.byref(vq), dist_th, int(chi2_gate), _p(best, C.c_int32), C.byref(ncmp))
    return best[:q.n], int(ncmp.value)


def search_init(f1, f2, prev_xy, window, ratio, check_ori):
    prev = np.ascontiguousarray(prev_xy, np.float32).copy()
    m12 = np.zeros(max(f1.n, 1), np.int32)
    ncmp = C.c_uint64(0)
    v1, v2 = f1.view(), f2.view()
    cnt = lib().orc_search_init(C.byref(v1), C.byref(v2), _p(prev, C.c_float), window, ratio, int(check_ori),
                                _p(m12, C.c_int32), C.byref(ncmp))
    return int(cnt), m12[:f1.n], prev, int(ncmp.value)


def stereo_match(left, right, n_rows, max_d):
    idx = np.zeros(max(left.n, 1), np.int32)
    dist = np.zeros(max(left.n, 1), np.int32)
    ncmp = C.c_uint64(0)
    vl, vr = left.view(), right.view()
    lib().orc_stereo_match(C.byref(vl), C.byref(vr), n_rows, max_d, _p(idx, C.c_int32), _p(dist, C.c_int32), C.byref(ncmp))
    return idx[:left.n], dist[:left.n], int(ncmp.value)


def distinctive_descriptor(desc) -> int:
    d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    return int(lib().orc_distinctive_descriptor(_p(d, C.c_uint8) if len(d) else None, len(d)))


def predict_scale(max_distance, current_dist, log_scale_factor, n_levels) -> int:
    return int(lib().orc_predict_scale(max_distance, current_dist, log_scale_factor, n_levels))


def transform(vocab, desc, levelsup: int, want_bow: bool = False):
    """-> word_id[n], node_id[n] (-1: weight not positive), (fv_node, fv_off, fv_idx) [, (bv_word, bv_value)]"""
    d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    n = len(d)
    word = np.zeros(max(n, 1), np.int32)
    node = np.zeros(max(n, 1), np.int32)
    fv_node = np.zeros(max(n, 1), np.uint32)
    fv_off = np.zeros(n + 1, np.uint32)
    fv_idx = np.zeros(max(n, 1), np.uint32)
    bv_n = C.c_int32(0)
    bv_word = np.zeros(max(n, 1), np.uint32)
    bv_value = np.zeros(max(n, 1), np.float64)
    v = vocab.view()
    nn = lib().orc_transform(C.byref(v), _p(d, C.c_uint8) if n else None, n, levelsup, _p(word, C.c_int32), _p(node, C.c_int32),
                             _p(fv_node, C.c_uint32), _p(fv_off, C.c_uint32), _p(fv_idx, C.c_uint32),
                             C.byref(bv_n), _p(bv_word, C.c_uint32), _p(bv_value, C.c_double))
    out = (word[:n], node[:n], (fv_node[:nn].copy(), fv_off[:nn + 1].copy(), fv_idx[:int(fv_off[nn])].copy()))
    if want_bow:
        k = int(bv_n.value)
        out = out + ((bv_word[:k].copy(), bv_value[:k].copy()),)
    return out


def bow_score(a, b):
    """-> (L1 score as DBoW2 computes it, number of common words)"""
    va, vb = a.view(), b.view()
    common = C.c_int32(0)
    s = lib().orc_bow_score(C.byref(va), C.byref(vb), C.byref(common))
    return float(s), int(common.value)


class _CKfDb(C.Structure):
    _fields_ = [("n_kf", C.c_int32), ("n_words", C.c_int32), ("bow", C.POINTER(CBowVecView)), ("mn_id", C.POINTER(C.c_int64)),
                ("nb_off", C.POINTER(C.c_int32)), ("nb_idx", C.POINTER(C.c_int32)), ("inv_off", C.POINTER(C.c_int32)),
                ("inv_idx", C.POINTER(C.c_int32)), ("query", C.POINTER(C.c_int64)), ("words", C.POINTER(C.c_int32)),
                ("score", C.POINTER(C.c_float))]


class KfDb:
    """A keyframe database in flat form for orc_detect_candidates: BowVectors, ids, the best-covisibility lists
    (GetBestCovisibilityKeyFrames(10), in order) and the inverted file (keyframes per word, in insertion order); query / words /
    score are the persistent mnLoopQuery|mnRelocQuery, mnLoopWords|mnRelocWords, mLoopScore|mRelocScore fields."""

    def __init__(self, bows, mn_id, neighbours, n_words):
        self.bows = list(bows)
        n = len(self.bows)
        self.mn_id = np.ascontiguousarray(mn_id, np.int64)
        self.nb_off = np.zeros(n + 1, np.int32)
        self.nb_off[1:] = np.cumsum([len(x) for x in neighbours])
        self.nb_idx = np.ascontiguousarray([k for x in neighbours for k in x] or [0], np.int32)
        inv = [[] for _ in range(n_words)]
        for k, b in enumerate(self.bows):                     # KeyFrameDatabase::add, KeyFrameDataBase.cc:53-60
            for w in b.word:
                inv[int(w)].append(k)
        self.inv_off = np.zeros(n_words + 1, np.int32)
        self.inv_off[1:] = np.cumsum([len(x) for x in inv])
        self.inv_idx = np.ascontiguousarray([k for x in inv for k in x] or [0], np.int32)
        self.query = np.full(n, -1, np.int64)
        self.words = np.zeros(n, np.int32)
        self.score = np.zeros(n, np.float32)
        self._views = (CBowVecView * max(n, 1))(*[b.view() for b in self.bows])
        self.n_words = n_words


def detect_candidates(is_reloc, db: KfDb, q_bow, q_id, connected=None, min_score=0.0):
    """KeyFrameDatabase::DetectRelocalizationCandidates / DetectLoopCandidates on a KfDb -> keyframe indices, in order."""
    L = lib()
    L.orc_detect_candidates.argtypes = [C.c_int, C.POINTER(_CKfDb), C.POINTER(CBowVecView), C.c_int64, C.POINTER(C.c_uint8), C.c_float,
                                        C.POINTER(C.c_int32)]
    n = len(db.bows)
    c = _CKfDb(n_kf=n, n_words=db.n_words, bow=db._views, mn_id=_p(db.mn_id, C.c_int64), nb_off=_p(db.nb_off, C.c_int32),
               nb_idx=_p(db.nb_idx, C.c_int32), inv_off=_p(db.inv_off, C.c_int32), inv_idx=_p(db.inv_idx, C.c_int32),
               query=_p(db.query, C.c_int64), words=_p(db.words, C.c_int32), score=_p(db.score, C.c_float))
    out = np.zeros(max(n, 1), np.int32)
    conn = None if connected is None else np.ascontiguousarray(connected, np.uint8)
    qv = q_bow.view()
    k = L.orc_detect_candidates(int(is_reloc), C.byref(c), C.byref(qv), int(q_id), _p(conn, C.c_uint8), float(min_score), _p(out, C.c_int32))
    return out[:k].tolist()
