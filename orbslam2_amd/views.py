"""ctypes mirrors of the flat views declared in include/orbmatch.h.

A `FrameData` (plain numpy arrays) is what the synthetic generator and the tests hold; `.view()`
produces the `orbm_frame_view` struct that both liborbmatch_hip.so and the CPU oracle take, and
keeps the arrays alive.  Field meaning and reference citations: include/orbmatch.h.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

MP_PRESENT, MP_BAD, MP_OBS = 1, 2, 4
Q_VALID, Q_OBS = 1, 2
GRID_ROWS, GRID_COLS = 48, 64
GRID_CELLS = GRID_ROWS * GRID_COLS
TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30

_u8p = C.POINTER(C.c_uint8)
_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)


class CFrameView(C.Structure):
    _fields_ = [
        ("n", C.c_int32),
        ("desc", _u8p),
        ("x", _f32p),
        ("y", _f32p),
        ("angle", _f32p),
        ("angle_raw", _f32p),
        ("octave", _i32p),
        ("uright", _f32p),
        ("mp_flags", _u8p),
        ("min_x", C.c_float),
        ("min_y", C.c_float),
        ("max_x", C.c_float),
        ("max_y", C.c_float),
        ("inv_w", C.c_float),
        ("inv_h", C.c_float),
        ("n_levels", C.c_int32),
        ("scale_factors", _f32p),
        ("level_sigma2", _f32p),
        ("n_nodes", C.c_int32),
        ("fv_node", _u32p),
        ("fv_off", _u32p),
        ("fv_idx", _u32p),
    ]


class CProjQueries(C.Structure):
    _fields_ = [
        ("n", C.c_int32),
        ("desc", _u8p),
        ("proj_x", _f32p),
        ("proj_y", _f32p),
        ("proj_xr", _f32p),
        ("level", _i32p),
        ("view_cos", _f32p),
        ("flags", _u8p),
    ]


class CWinQueries(C.Structure):
    _fields_ = [
        ("n", C.c_int32),
        ("desc", _u8p),
        ("u", _f32p),
        ("v", _f32p),
        ("radius", _f32p),
        ("min_level", _i32p),
        ("max_level", _i32p),
        ("ur", _f32p),
        ("angle", _f32p),
        ("flags", _u8p),
    ]


class CVocabView(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int32),
        ("depth", C.c_int32),
        ("child_off", _u32p),
        ("child_idx", _u32p),
        ("desc", _u8p),
        ("word_id", _i32p),
        ("weight", C.POINTER(C.c_double)),
        ("weighting", C.c_int32),
        ("scoring", C.c_int32),
    ]


class CBowVecView(C.Structure):
    _fields_ = [("n", C.c_int32), ("word", _u32p), ("value", C.POINTER(C.c_double))]


def _ptr(a: Optional[np.ndarray], ctype):
    if a is None:
        return C.cast(None, C.POINTER(ctype))
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(ctype))


def _as(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


def scale_tables(n_levels: int = 8, scale_factor: float = 1.2):
    """mvScaleFactors / mvLevelSigma2 exactly as ORBextractor.cc:466-470 builds them (float32)."""
    sf = np.ones(n_levels, dtype=np.float32)
    s2 = np.ones(n_levels, dtype=np.float32)
    f = np.float32(scale_factor)
    for i in range(1, n_levels):
        sf[i] = np.float32(sf[i - 1] * f)
        s2[i] = np.float32(sf[i] * sf[i])
    return sf, s2


@dataclass
class FrameData:
    desc: np.ndarray                      # [n,32] uint8
    x: np.ndarray                         # [n] float32 (mvKeysUn)
    y: np.ndarray
    angle: np.ndarray                     # [n] float32 degrees
    octave: np.ndarray                    # [n] int32
    uright: Optional[np.ndarray] = None   # [n] float32 or None (mono)
    mp_flags: Optional[np.ndarray] = None # [n] uint8 or None
    angle_raw: Optional[np.ndarray] = None
    bounds: tuple = (0.0, 0.0, 640.0, 480.0)  # min_x, min_y, max_x, max_y
    scale_factors: np.ndarray = None
    level_sigma2: np.ndarray = None
    fv_node: Optional[np.ndarray] = None  # [n_nodes] uint32 ascending
    fv_off: Optional[np.ndarray] = None   # [n_nodes+1] uint32
    fv_idx: Optional[np.ndarray] = None   # [n] uint32
    _keep: list = field(default_factory=list, repr=False)

    def __post_init__(self):
        self.desc = _as(self.desc, np.uint8).reshape(-1, 32)
        n = self.desc.shape[0]
        self.x = _as(self.x, np.float32)
        self.y = _as(self.y, np.float32)
        self.angle = _as(self.angle, np.float32)
        self.octave = _as(self.octave, np.int32)
        self.uright = _as(self.uright, np.float32)
        self.mp_flags = _as(self.mp_flags, np.uint8)
        self.angle_raw = _as(self.angle_raw, np.float32)
        if self.scale_factors is None:
            self.scale_factors, self.level_sigma2 = scale_tables()
        self.scale_factors = _as(self.scale_factors, np.float32)
        self.level_sigma2 = _as(self.level_sigma2, np.float32)
        if self.fv_node is None:
            self.fv_node = np.zeros(0, np.uint32)
            self.fv_off = np.zeros(1, np.uint32)
            self.fv_idx = np.zeros(0, np.uint32)
        self.fv_node = _as(self.fv_node, np.uint32)
        self.fv_off = _as(self.fv_off, np.uint32)
        self.fv_idx = _as(self.fv_idx, np.uint32)
        for a in (self.x, self.y, self.angle, self.octave):
            assert a.shape == (n,), (a.shape, n)

    @property
    def n(self) -> int:
        return int(self.desc.shape[0])

    def view(self) -> CFrameView:
        min_x, min_y, max_x, max_y = (np.float32(b) for b in self.bounds)
        # Frame.cc:255-256: static_cast<float>(FRAME_GRID_COLS)/static_cast<float>(mnMaxX-mnMinX)
        inv_w = np.float32(np.float32(GRID_COLS) / np.float32(max_x - min_x))
        inv_h = np.float32(np.float32(GRID_ROWS) / np.float32(max_y - min_y))
        v = CFrameView(
            n=self.n, desc=_ptr(self.desc, C.c_uint8), x=_ptr(self.x, C.c_float), y=_ptr(self.y, C.c_float),
            angle=_ptr(self.angle, C.c_float), angle_raw=_ptr(self.angle_raw, C.c_float),
            octave=_ptr(self.octave, C.c_int32), uright=_ptr(self.uright, C.c_float),
            mp_flags=_ptr(self.mp_flags, C.c_uint8),
            min_x=float(min_x), min_y=float(min_y), max_x=float(max_x), max_y=float(max_y),
            inv_w=float(inv_w), inv_h=float(inv_h),
            n_levels=int(self.scale_factors.shape[0]),
            scale_factors=_ptr(self.scale_factors, C.c_float), level_sigma2=_ptr(self.level_sigma2, C.c_float),
            n_nodes=int(self.fv_node.shape[0]), fv_node=_ptr(self.fv_node, C.c_uint32),
            fv_off=_ptr(self.fv_off, C.c_uint32), fv_idx=_ptr(self.fv_idx, C.c_uint32))
        v._owner = self  # keep arrays alive as long as the struct lives
        return v


@dataclass
class ProjQueries:
    desc: np.ndarray
    proj_x: np.ndarray
    proj_y: np.ndarray
    proj_xr: np.ndarray
    level: np.ndarray
    view_cos: np.ndarray
    flags: Optional[np.ndarray] = None

    def __post_init__(self):
        self.desc = _as(self.desc, np.uint8).reshape(-1, 32)
        self.proj_x = _as(self.proj_x, np.float32)
        self.proj_y = _as(self.proj_y, np.float32)
        self.proj_xr = _as(self.proj_xr, np.float32)
        self.level = _as(self.level, np.int32)
        self.view_cos = _as(self.view_cos, np.float32)
        self.flags = _as(self.flags, np.uint8)

    @property
    def n(self) -> int:
        return int(self.desc.shape[0])

    def view(self) -> CProjQueries:
        v = CProjQueries(n=self.n, desc=_ptr(self.desc, C.c_uint8), proj_x=_ptr(self.proj_x, C.c_float),
                         proj_y=_ptr(self.proj_y, C.c_float), proj_xr=_ptr(self.proj_xr, C.c_float),
                         level=_ptr(self.level, C.c_int32), view_cos=_ptr(self.view_cos, C.c_float),
                         flags=_ptr(self.flags, C.c_uint8))
        v._owner = self
        return v


@dataclass
class WinQueries:
    desc: np.ndarray
    u: np.ndarray
    v: np.ndarray
    radius: np.ndarray
    min_level: np.ndarray
    max_level: np.ndarray
    ur: Optional[np.ndarray] = None
    angle: Optional[np.ndarray] = None
    flags: Optional[np.ndarray] = None

    def __post_init__(self):
        self.desc = _as(self.desc, np.uint8).reshape(-1, 32)
        self.u = _as(self.u, np.float32)
        self.v = _as(self.v, np.float32)
        self.radius = _as(self.radius, np.float32)
        self.min_level = _as(self.min_level, np.int32)
        self.max_level = _as(self.max_level, np.int32)
        self.ur = _as(self.ur, np.float32)
        self.angle = _as(self.angle, np.float32)
        self.flags = _as(self.flags, np.uint8)

    @property
    def n(self) -> int:
        return int(self.desc.shape[0])

    def view(self) -> CWinQueries:
        w = CWinQueries(n=self.n, desc=_ptr(self.desc, C.c_uint8), u=_ptr(self.u, C.c_float),
                        v=_ptr(self.v, C.c_float), radius=_ptr(self.radius, C.c_float),
                        min_level=_ptr(self.min_level, C.c_int32), max_level=_ptr(self.max_level, C.c_int32),
                        ur=_ptr(self.ur, C.c_float), angle=_ptr(self.angle, C.c_float),
                        flags=_ptr(self.flags, C.c_uint8))
        w._owner = self
        return w


@dataclass
class VocabData:
    """Flat DBoW2 vocabulary tree (orbm_vocab_view): node 0 = root, children of node i = child_idx[child_off[i]:child_off[i+1]]."""
    depth: int
    child_off: np.ndarray   # [n_nodes+1] uint32
    child_idx: np.ndarray   # uint32
    desc: np.ndarray        # [n_nodes,32] uint8
    word_id: np.ndarray     # [n_nodes] int32, -1 for inner nodes
    weight: Optional[np.ndarray] = None   # [n_nodes] float64
    weighting: int = 0                    # DBoW2 WeightingType (0 = TF_IDF)
    scoring: int = 0                      # DBoW2 ScoringType (0 = L1_NORM)

    def __post_init__(self):
        self.child_off = _as(self.child_off, np.uint32)
        self.child_idx = _as(self.child_idx, np.uint32)
        self.desc = _as(self.desc, np.uint8).reshape(-1, 32)
        self.word_id = _as(self.word_id, np.int32)
        self.weight = _as(self.weight, np.float64)

    @property
    def n_nodes(self) -> int:
        return int(self.desc.shape[0])

    def view(self) -> CVocabView:
        v = CVocabView(n_nodes=self.n_nodes, depth=int(self.depth), child_off=_ptr(self.child_off, C.c_uint32),
                       child_idx=_ptr(self.child_idx, C.c_uint32), desc=_ptr(self.desc, C.c_uint8),
                       word_id=_ptr(self.word_id, C.c_int32), weight=_ptr(self.weight, C.c_double),
                       weighting=int(self.weighting), scoring=int(self.scoring))
        v._owner = self
        return v


@dataclass
class BowVecData:
    """Flat DBoW2::BowVector (orbm_bowvec_view): ascending word ids and their values."""
    word: np.ndarray    # [n] uint32
    value: np.ndarray   # [n] float64

    def __post_init__(self):
        self.word = _as(self.word, np.uint32)
        self.value = _as(self.value, np.float64)
        assert self.word.shape == self.value.shape

    @property
    def n(self) -> int:
        return int(self.word.shape[0])

    def view(self) -> CBowVecView:
        v = CBowVecView(n=self.n, word=_ptr(self.word, C.c_uint32), value=_ptr(self.value, C.c_double))
        v._owner = self
        return v
