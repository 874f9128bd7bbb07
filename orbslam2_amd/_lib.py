"""Loader of liborbmatch_hip.so (the C ABI of include/orbmatch.h).  There is no CPU fallback: if the
library is missing or fails to load, importing a matcher raises."""
import ctypes as C
import os

from .views import CBowVecView, CFrameView, CProjQueries, CVocabView, CWinQueries

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liborbmatch_hip.so")
# scripts/ only: ORBM_DIAG_LIB=1 loads the build with the timing diagnostics compiled in (`make -C orbslam2_amd/csrc diag`)
if os.environ.get("ORBM_DIAG_LIB") == "1":
    LIB_PATH = os.path.join(_HERE, "liborbmatch_hip_diag.so")

# every symbol include/orbmatch.h declares
SYMBOLS = [
    "orbm_create", "orbm_destroy", "orbm_last_error", "orbm_set_stream", "orbm_synchronize", "orbm_device_info",
    "orbm_kernel_timing", "orbm_kernel_times",
    "orbm_frames_upload", "orbm_frames_free", "orbm_frame_set_mp_flags", "orbm_frame_grid_download",
    "orbm_proj_queries_upload", "orbm_win_queries_upload", "orbm_queries_free",
    "orbm_hamming256_pairs", "orbm_distinctive_descriptors_batch", "orbm_stereo_match_batch", "orbm_bf2nn_batch", "orbm_search_bow_batch", "orbm_search_triang_batch",
    "orbm_search_proj_points_batch", "orbm_search_window_batch", "orbm_search_window_free_batch", "orbm_search_window_free_once", "orbm_search_init",
    "orbm_vocab_upload", "orbm_vocab_free", "orbm_frames_compute_bow", "orbm_vocab_transform", "orbm_frame_fv_download",
    "orbm_bowvecs_upload", "orbm_bowvecs_free", "orbm_bow_score_batch", "orbm_bowvec_download",
    "orbm_popcount_peak", "orbm_search_bow_batch_noalloc",
    "orbm_bf2nn_prepare", "orbm_pack_records", "orbm_allpairs_batch", "orbm_allpairs_shard", "orbm_allpairs", "orbm_search_proj_points_once", "orbm_search_window_once",
]

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `make -C orbslam2_amd/csrc` (hipcc, gfx950). "
            "orbslam2_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    i32, f32 = C.c_int32, C.c_float
    L.orbm_create.restype = vp
    L.orbm_create.argtypes = [C.c_int]
    L.orbm_destroy.restype = None
    L.orbm_destroy.argtypes = [vp]
    L.orbm_last_error.restype = C.c_char_p
    L.orbm_last_error.argtypes = [vp]
    L.orbm_set_stream.argtypes = [vp, vp]
    L.orbm_synchronize.argtypes = [vp]
    L.orbm_kernel_timing.argtypes = [vp, C.c_int]
    L.orbm_kernel_times.argtypes = [vp, vp, C.c_int]
    L.orbm_device_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.c_char_p, C.c_int]
    L.orbm_frames_upload.argtypes = [vp, C.c_int, C.POINTER(CFrameView), vp]
    L.orbm_frames_free.argtypes = [vp, C.c_int, vp]
    L.orbm_frame_set_mp_flags.argtypes = [vp, i32, vp]
    L.orbm_frame_grid_download.argtypes = [vp, i32, vp, vp]
    L.orbm_proj_queries_upload.argtypes = [vp, C.c_int, C.POINTER(CProjQueries), vp]
    L.orbm_win_queries_upload.argtypes = [vp, C.c_int, C.POINTER(CWinQueries), vp]
    L.orbm_queries_free.argtypes = [vp, C.c_int, vp]
    L.orbm_hamming256_pairs.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int]
    L.orbm_distinctive_descriptors_batch.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int]
    L.orbm_stereo_match_batch.argtypes = [vp, C.c_int, vp, vp, vp, i32, vp, vp, C.c_int]
    L.orbm_bf2nn_batch.argtypes = [vp, C.c_int, vp, vp, f32, C.c_int, i32, vp, vp, vp, vp, C.c_int]
    L.orbm_search_bow_batch.argtypes = [vp, C.c_int, C.c_int, vp, vp, f32, C.c_int, i32, vp, vp, C.c_int]
    L.orbm_search_bow_batch_noalloc.argtypes = [vp, C.c_int, C.c_int, vp, vp, f32, C.c_int, i32, vp, vp]
    L.orbm_bf2nn_prepare.argtypes = [vp]
    L.orbm_pack_records.argtypes = [vp, C.c_int, vp, vp, vp, i32, C.c_int, vp, vp, C.c_uint32, vp, C.c_int]
    L.orbm_allpairs_batch.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, f32, C.c_int, C.c_int, i32, C.c_int, vp, vp, C.c_uint32, vp, C.c_int]
    L.orbm_allpairs.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, f32, C.c_int, C.c_int, i32, C.c_int, vp, vp, C.c_uint32, vp]
    L.orbm_search_proj_points_once.argtypes = [vp, i32, C.POINTER(CProjQueries), f32, f32, vp, vp, vp]
    L.orbm_search_window_once.argtypes = [vp, C.c_int, i32, C.POINTER(CWinQueries), C.c_int, C.c_int, vp, vp, vp]
    L.orbm_allpairs_shard.restype = C.c_int64
    L.orbm_allpairs_shard.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, vp, vp]
    L.orbm_search_triang_batch.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, i32, vp, vp, C.c_int]
    L.orbm_search_proj_points_batch.argtypes = [vp, C.c_int, vp, vp, f32, f32, vp, i32, vp, vp, C.c_int]
    L.orbm_search_window_batch.argtypes = [vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, vp, i32, vp, vp, C.c_int]
    L.orbm_search_window_free_batch.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, i32, vp, C.c_int]
    L.orbm_search_window_free_once.argtypes = [vp, C.c_int, vp, C.POINTER(CWinQueries), C.c_int, C.c_int, i32, vp]
    L.orbm_search_init.argtypes = [vp, i32, i32, vp, C.c_int, f32, C.c_int, vp]
    L.orbm_vocab_upload.argtypes = [vp, C.POINTER(CVocabView), C.POINTER(i32)]
    L.orbm_vocab_free.argtypes = [vp, i32]
    L.orbm_frames_compute_bow.argtypes = [vp, i32, C.c_int, vp, C.c_int, i32, vp, vp, vp]
    L.orbm_vocab_transform.argtypes = [vp, i32, C.c_int, vp, C.c_int, vp, vp, C.c_int]
    L.orbm_frame_fv_download.argtypes = [vp, i32, C.POINTER(i32), vp, vp, vp]
    L.orbm_bowvecs_upload.argtypes = [vp, C.c_int, C.POINTER(CBowVecView), vp]
    L.orbm_bowvecs_free.argtypes = [vp, C.c_int, vp]
    L.orbm_bowvec_download.argtypes = [vp, i32, C.POINTER(i32), vp, vp, i32]
    L.orbm_bow_score_batch.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_int]
    L.orbm_popcount_peak.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
    for name in SYMBOLS:
        getattr(L, name)  # raises AttributeError if the library does not export a declared symbol
    _lib = L
    return L
