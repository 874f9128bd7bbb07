"""Seeded synthetic inputs for the matcher hot path.

There is no dataset, vocabulary or image in the reference tree, so every test and benchmark input
is generated here: 256-bit landmark descriptors, noisy observations (bit flips), level-major
keypoints with the per-level counts of ORBextractor.cc:486-509, FeatureVectors whose node id is a
function of the landmark, and projected map points.  Everything is numpy, deterministic per seed.
"""
from __future__ import annotations

import numpy as np

from .views import (FrameData, ProjQueries, WinQueries, MP_PRESENT, MP_OBS, MP_BAD, Q_VALID, Q_OBS,
                    scale_tables)

SEED0 = 20161011  # seed = SEED0 + BASELINE config number (1-based)


def level_counts(n: int, n_levels: int = 8, scale_factor: float = 1.2) -> np.ndarray:
    """mnFeaturesPerLevel (ORBextractor.cc:486-509)."""
    factor = np.float32(1.0) / np.float32(scale_factor)
    desired = np.float32(n) * (np.float32(1) - factor) / (
        np.float32(1) - np.float32(np.power(np.float64(factor), np.float64(n_levels))))
    counts = np.zeros(n_levels, dtype=np.int64)
    s = 0
    for lvl in range(n_levels - 1):
        counts[lvl] = int(np.rint(desired))  # cvRound: round half to even
        s += counts[lvl]
        desired = np.float32(desired * factor)
    counts[n_levels - 1] = max(n - s, 0)
    # rounding can overshoot n by a few; trim from the top like a capped extractor would
    over = counts.sum() - n
    lvl = n_levels - 2
    while over > 0 and lvl >= 0:
        d = min(over, counts[lvl])
        counts[lvl] -= d
        over -= d
        lvl -= 1
    return counts


def octaves_level_major(n: int, n_levels: int = 8) -> np.ndarray:
    return np.repeat(np.arange(n_levels, dtype=np.int32), level_counts(n, n_levels))[:n]


def random_descriptors(rng: np.random.Generator, n: int) -> np.ndarray:
    return rng.integers(0, 256, size=(n, 32), dtype=np.uint8)


def observe(rng: np.random.Generator, desc: np.ndarray, p_flip: float = 0.08) -> np.ndarray:
    """Noisy observation: every bit flipped with probability p_flip."""
    flips = rng.random((desc.shape[0], 256)) < p_flip
    return desc ^ np.packbits(flips, axis=1, bitorder="little")


def node_of_landmark(lm_id: np.ndarray, n_nodes: int = 100) -> np.ndarray:
    """Deterministic landmark -> vocabulary node (stands in for DBoW2 transform(...,4), Frame.cc:531).
    Node ids are sparse (not 0..n-1) so that the merge walk's lower_bound skips are exercised."""
    h = (lm_id.astype(np.uint64) * np.uint64(2654435761)) >> np.uint64(7)
    return (np.uint64(1000) + np.uint64(13) * (h % np.uint64(n_nodes))).astype(np.uint32)


def feature_vector(node: np.ndarray):
    """CSR of an ordered map node -> [feature indices in ascending order]."""
    order = np.argsort(node, kind="stable")
    nodes, counts = np.unique(node[order], return_counts=True)
    off = np.zeros(len(nodes) + 1, dtype=np.uint32)
    off[1:] = np.cumsum(counts)
    return nodes.astype(np.uint32), off, order.astype(np.uint32)


def make_frame(rng, lm_desc: np.ndarray, lm_id: np.ndarray, *, x=None, y=None, angle=None, octave=None,
               p_flip=0.08, n_nodes=100, node_noise=0.1, mp_flags=None, uright=None,
               bounds=(0.0, 0.0, 640.0, 480.0), with_fv=True) -> FrameData:
    n = len(lm_id)
    desc = observe(rng, lm_desc[lm_id], p_flip)
    if octave is None:
        octave = octaves_level_major(n)
    if x is None:
        x = rng.uniform(16, 624, n).astype(np.float32)
    if y is None:
        y = rng.uniform(16, 464, n).astype(np.float32)
    if angle is None:
        angle = rng.uniform(0, 360, n).astype(np.float32)
    kw = {}
    if with_fv:
        node = node_of_landmark(lm_id, n_nodes)
        noisy = rng.random(n) < node_noise
        node = np.where(noisy, node_of_landmark(rng.integers(0, 1 << 30, n), n_nodes), node)
        fv_node, fv_off, fv_idx = feature_vector(node)
        kw = dict(fv_node=fv_node, fv_off=fv_off, fv_idx=fv_idx)
    return FrameData(desc=desc, x=x, y=y, angle=np.mod(angle, 360).astype(np.float32), octave=octave,
                     uright=uright, mp_flags=mp_flags, bounds=bounds, **kw)


def _paired_frame(rng, f1: FrameData, lm_desc, lm1, n_pool, share, disp_sigma, rot_deg, rot_sigma, **kw):
    """Second view of f1: a fraction `share` of keypoints re-observe f1's landmark (same octave,
    displaced position, angle offset so that angle1-angle2 ~ rot_deg); the rest see new landmarks."""
    n = f1.n
    shared = rng.random(n) < share
    lm2 = np.where(shared, lm1, rng.integers(n_pool, 2 * n_pool, n))
    x2 = np.where(shared, f1.x + rng.normal(0, disp_sigma, n), rng.uniform(16, 624, n)).astype(np.float32)
    y2 = np.where(shared, f1.y + rng.normal(0, disp_sigma, n), rng.uniform(16, 464, n)).astype(np.float32)
    x2 = np.clip(x2, 1, 638).astype(np.float32)
    y2 = np.clip(y2, 1, 478).astype(np.float32)
    ang2 = np.where(shared, f1.angle - rot_deg + rng.normal(0, rot_sigma, n), rng.uniform(0, 360, n))
    # shuffle within each octave so that index order differs between the views
    perm = np.arange(n)
    for lvl in np.unique(f1.octave):
        idx = np.nonzero(f1.octave == lvl)[0]
        perm[idx] = rng.permutation(idx)
    f2 = make_frame(rng, lm_desc, lm2[perm], x=x2[perm], y=y2[perm], angle=ang2[perm],
                    octave=f1.octave[perm], **kw)
    return f2, perm


def init_pair(seed=SEED0 + 1, n=1000):
    """BASELINE config 1: two 640x480 frames, 60% shared landmarks, N(0,15px) displacement."""
    rng = np.random.default_rng(seed)
    pool = 2 * n
    lm_desc = random_descriptors(rng, 2 * pool)
    lm1 = rng.permutation(pool)[:n]
    f1 = make_frame(rng, lm_desc, lm1, with_fv=False)
    f2, _ = _paired_frame(rng, f1, lm_desc, lm1, pool, 0.6, 15.0, 7.0, 3.0, with_fv=False)
    prev_xy = np.stack([f1.x, f1.y], axis=1).astype(np.float32).copy()
    return f1, f2, prev_xy


def bf_pair(seed=SEED0 + 2, n=2000, share=0.5):
    """BASELINE config 2: 2000x2000 brute force, 50% shared landmarks, p_flip 0.08."""
    rng = np.random.default_rng(seed)
    pool = 2 * n
    lm_desc = random_descriptors(rng, 2 * pool)
    lm1 = rng.permutation(pool)[:n]
    f1 = make_frame(rng, lm_desc, lm1, with_fv=False)
    f2, _ = _paired_frame(rng, f1, lm_desc, lm1, pool, share, 15.0, 7.0, 3.0, with_fv=False)
    return f1, f2


def keyframe_pool(seed=SEED0 + 5, n_kf=64, n=2000, window=None, mp_valid_frac=1.0, stereo_frac=0.0,
                  bad_frac=0.0):
    """Keyframes along a synthetic trajectory: keyframe k observes n landmarks drawn from a sliding
    window of the landmark pool, so neighbours overlap ~60% and far keyframes ~0% (config 5);
    window=None makes every keyframe draw from one common pool (config 4 style, ~30% overlap)."""
    rng = np.random.default_rng(seed)
    if window is None:
        pool = int(n / 0.3)
        starts = np.zeros(n_kf, dtype=np.int64)
        width = pool
    else:
        width = window
        step = max(1, int(width * 0.25))
        starts = np.arange(n_kf, dtype=np.int64) * step
        pool = int(starts[-1] + width)
    lm_desc = random_descriptors(rng, pool)
    lm_angle = rng.uniform(0, 360, pool)
    lm_oct = rng.integers(0, 8, pool)
    frames = []
    for k in range(n_kf):
        lm = starts[k] + rng.permutation(width)[:n]
        # keep the level-major layout: sort by the landmark's octave
        octv = lm_oct[lm].astype(np.int32)
        order = np.argsort(octv, kind="stable")
        lm, octv = lm[order], octv[order]
        rot = rng.uniform(0, 360)
        ang = lm_angle[lm] + rot + rng.normal(0, 3.0, n)
        flags = np.zeros(n, np.uint8)
        has = rng.random(n) < mp_valid_frac
        flags[has] = MP_PRESENT | MP_OBS
        flags[has & (rng.random(n) < bad_frac)] |= MP_BAD
        uright = None
        if stereo_frac > 0:
            uright = np.where(rng.random(n) < stereo_frac, rng.uniform(1, 600, n), -1.0).astype(np.float32)
        frames.append(make_frame(rng, lm_desc, lm, angle=ang, octave=octv, mp_flags=flags, uright=uright))
    return frames


def proj_instance(seed=SEED0 + 3, n=2000, n_queries=50000, th_noise=2.0, stereo=False, claimed_frac=0.0,
                  obs_frac=1.0, valid_frac=1.0):
    """BASELINE config 3: a 2000-keypoint frame and n_queries projected map points; each query is a
    noisy re-observation of one frame keypoint's landmark, in random order (local map order)."""
    rng = np.random.default_rng(seed)
    lm_desc = random_descriptors(rng, n)
    flags = np.zeros(n, np.uint8)
    cl = rng.random(n) < claimed_frac
    flags[cl] = MP_PRESENT | MP_OBS
    flags[cl & (rng.random(n) < 0.3)] = MP_PRESENT  # present but Observations()==0: still matchable
    uright = None
    if stereo:
        uright = np.where(rng.random(n) < 0.7, rng.uniform(1, 600, n), -1.0).astype(np.float32)
    f = make_frame(rng, lm_desc, np.arange(n), with_fv=False, mp_flags=flags, uright=uright)
    src = rng.integers(0, n, n_queries)
    qdesc = observe(rng, lm_desc[src], 0.08)
    px = (f.x[src] + rng.normal(0, th_noise, n_queries)).astype(np.float32)
    py = (f.y[src] + rng.normal(0, th_noise, n_queries)).astype(np.float32)
    level = np.clip(f.octave[src] + (rng.random(n_queries) < 0.3), 0, 7).astype(np.int32)
    view_cos = rng.uniform(0.9, 1.0, n_queries).astype(np.float32)
    if stereo:
        pxr = np.where(f.uright[src] > 0, f.uright[src] + rng.normal(0, 3.0, n_queries), -1.0).astype(np.float32)
    else:
        pxr = np.full(n_queries, -1.0, np.float32)
    qflags = np.zeros(n_queries, np.uint8)
    qflags[rng.random(n_queries) < valid_frac] |= Q_VALID
    qflags[rng.random(n_queries) < obs_frac] |= Q_OBS
    q = ProjQueries(desc=qdesc, proj_x=px, proj_y=py, proj_xr=pxr, level=level, view_cos=view_cos, flags=qflags)
    return f, q


def window_instance(seed, mode, n=2000, n_queries=1500, th=7.0, stereo=False, claimed_frac=0.1):
    """Queries of the best-1 windowed overloads (orbm_search_window_batch modes 0..2) in the flat form the
    host side produces after projection: centre, radius, level range, optional ur and angle."""
    rng = np.random.default_rng(seed)
    lm_desc = random_descriptors(rng, n)
    flags = np.zeros(n, np.uint8)
    cl = rng.random(n) < claimed_frac
    flags[cl] = MP_PRESENT | MP_OBS
    if mode == 0:
        flags[cl & (rng.random(n) < 0.3)] = MP_PRESENT
    uright = None
    if stereo:
        uright = np.where(rng.random(n) < 0.7, rng.uniform(1, 600, n), -1.0).astype(np.float32)
    bounds = (0.0, 0.0, 640.0, 480.0)
    f = make_frame(rng, lm_desc, np.arange(n), with_fv=False, mp_flags=flags, uright=uright, bounds=bounds)
    src = rng.integers(0, n, n_queries)
    qdesc = observe(rng, lm_desc[src], 0.08)
    u = (f.x[src] + rng.normal(0, 3.0, n_queries)).astype(np.float32)
    v = (f.y[src] + rng.normal(0, 3.0, n_queries)).astype(np.float32)
    lvl = np.clip(f.octave[src] + rng.integers(-1, 2, n_queries), 0, 7).astype(np.int32)
    sf, _ = scale_tables()
    radius = (np.float32(th) * sf[lvl]).astype(np.float32)
    if mode == 0:
        kind = rng.integers(0, 3, n_queries)  # forward / backward / default level ranges (:1859-1864)
        min_level = np.where(kind == 0, lvl, np.where(kind == 1, 0, lvl - 1)).astype(np.int32)
        max_level = np.where(kind == 0, -1, np.where(kind == 1, lvl, lvl + 1)).astype(np.int32)
    elif mode == 1:
        min_level, max_level = (lvl - 1).astype(np.int32), (lvl + 1).astype(np.int32)
    else:
        min_level, max_level = (lvl - 1).astype(np.int32), lvl.astype(np.int32)
    ur = None
    if stereo and mode == 0:
        ur = np.where(f.uright[src] > 0, f.uright[src] + rng.normal(0, 4.0, n_queries), -5.0).astype(np.float32)
    rot = 11.0
    angle = np.mod(f.angle[src] + rot + rng.normal(0, 3.0, n_queries), 360).astype(np.float32)
    qflags = np.zeros(n_queries, np.uint8)
    qflags[rng.random(n_queries) < 0.9] |= Q_VALID
    qflags[rng.random(n_queries) < 0.8] |= Q_OBS
    q = WinQueries(desc=qdesc, u=u, v=v, radius=radius, min_level=min_level, max_level=max_level, ur=ur,
                   angle=None if mode == 2 else angle, flags=qflags)
    return f, q


def triang_pair(seed, n=2000, stereo_frac=0.3, has_mp_frac=0.5):
    """Two keyframes for SearchForTriangulation plus a plausible F12 and epipole: the second view is a
    small horizontal shift of the first so that true matches lie near their epipolar lines."""
    rng = np.random.default_rng(seed)
    pool = 2 * n
    lm_desc = random_descriptors(rng, 2 * pool)
    lm1 = rng.permutation(pool)[:n]

    def flags_and_ur():
        fl = np.where(rng.random(n) < has_mp_frac, MP_PRESENT | MP_OBS, 0).astype(np.uint8)
        ur = np.where(rng.random(n) < stereo_frac, rng.uniform(1, 600, n), -1.0).astype(np.float32)
        return fl, ur

    fl1, ur1 = flags_and_ur()
    f1 = make_frame(rng, lm_desc, lm1, mp_flags=fl1, uright=ur1)
    fl2, ur2 = flags_and_ur()
    f2, _ = _paired_frame(rng, f1, lm_desc, lm1, pool, 0.6, 1.0, 4.0, 3.0, mp_flags=fl2, uright=ur2)
    # pure x-translation between the cameras: epipolar lines are horizontal (y2 = y1): F = [t]_x
    # with t=(1,0,0) -> l = x1^T F = (0, -1, y1)  => row-major F12 = [[0,0,0],[0,0,1],[0,-1,0]]
    f12 = np.array([0, 0, 0, 0, 0, 1, 0, -1, 0], dtype=np.float32)
    f12 = (f12 + rng.normal(0, 2e-6, 9)).astype(np.float32)
    epipole = np.array([320.0 + rng.normal(0, 50), 240.0 + rng.normal(0, 50)], dtype=np.float32)
    return f1, f2, f12, epipole


def make_vocabulary(seed=SEED0 + 9, k=10, depth=3, irregular=False, flip=0.2, tie_frac=0.0, stop_frac=0.0):
    """Synthetic DBoW2-shaped vocabulary tree, breadth-first node ids (the order DBoW2 saves and loads), children of a
    node = noisy copies of it (flip = per-bit flip probability) so that descents are decisive at the top and close at
    the bottom.  irregular: 1..k children per node and leaves above the last level (k-means clusters that could not be
    split).  tie_frac: fraction of nodes whose descriptor is copied from their elder sibling (exact distance ties: the
    first child must win).  stop_frac: fraction of words with weight 0 (left out of the FeatureVector)."""
    from .views import VocabData
    rng = np.random.default_rng(seed)
    desc = [np.zeros(32, np.uint8)]
    level = [0]
    children = [[]]
    frontier = [0]
    for lv in range(1, depth + 1):
        nxt = []
        for parent in frontier:
            if irregular and lv > 1 and rng.random() < 0.15:
                continue                                     # stays a leaf above the last level
            nc = int(rng.integers(1, k + 1)) if irregular else k
            base = desc[parent] if lv > 1 else None
            for c in range(nc):
                if base is None:
                    d = rng.integers(0, 256, 32, dtype=np.uint8)
                else:
                    bits = np.unpackbits(base)
                    bits ^= (rng.random(256) < flip).astype(np.uint8)
                    d = np.packbits(bits)
                if c > 0 and rng.random() < tie_frac:
                    d = desc[children[parent][c - 1]].copy()
                nid = len(desc)
                desc.append(d)
                level.append(lv)
                children.append([])
                children[parent].append(nid)
                nxt.append(nid)
        frontier = nxt
    n_nodes = len(desc)
    child_off = np.zeros(n_nodes + 1, np.uint32)
    child_idx = []
    word_id = np.full(n_nodes, -1, np.int32)
    nw = 0
    for i in range(n_nodes):
        child_off[i + 1] = child_off[i] + len(children[i])
        child_idx.extend(children[i])
        if not children[i]:
            word_id[i] = nw
            nw += 1
    weight = rng.random(n_nodes) * 5 + 0.01
    weight[rng.random(n_nodes) < stop_frac] = 0.0
    return VocabData(depth=depth, child_off=child_off, child_idx=np.asarray(child_idx, np.uint32), desc=np.stack(desc),
                     word_id=word_id, weight=weight)


def vocabulary_features(rng, vocab, n, flip=0.1):
    """n descriptors near random nodes of the tree (so descents meet close calls), plus a few exact node descriptors."""
    pick = rng.integers(1, vocab.n_nodes, n)
    bits = np.unpackbits(vocab.desc[pick], axis=1)
    noisy = bits ^ (rng.random(bits.shape) < flip).astype(np.uint8)
    noisy[: n // 8] = bits[: n // 8]
    return np.packbits(noisy, axis=1)


def make_vocabulary_regular(seed=SEED0 + 10, k=10, depth=6):
    """Full k-ary tree of the given depth (ORBvoc.txt is k = 10, depth = 6: 1 111 111 nodes, 10^6 words), breadth-first ids,
    built level by level with array operations: a child = its parent with about 1/8 of the bits flipped (1/4 below the
    root's children)."""
    from .views import VocabData
    rng = np.random.default_rng(seed)
    levels = [np.zeros((1, 32), np.uint8)]
    for lv in range(1, depth + 1):
        parent = np.repeat(levels[-1], k, axis=0)
        m = parent.shape[0]
        if lv == 1:
            child = rng.integers(0, 256, (m, 32), dtype=np.uint8)
        else:
            flips = rng.integers(0, 256, (m, 32), dtype=np.uint8) & rng.integers(0, 256, (m, 32), dtype=np.uint8)
            if lv > 2:
                flips &= rng.integers(0, 256, (m, 32), dtype=np.uint8)
            child = parent ^ flips
        levels.append(child)
    desc = np.concatenate(levels)
    n_nodes = desc.shape[0]
    n_inner = n_nodes - levels[-1].shape[0]
    child_off = np.minimum(np.arange(n_nodes + 1, dtype=np.int64), n_inner) * k
    child_idx = np.arange(1, n_nodes, dtype=np.uint32)
    word_id = np.full(n_nodes, -1, np.int32)
    word_id[n_inner:] = np.arange(n_nodes - n_inner)
    return VocabData(depth=depth, child_off=child_off.astype(np.uint32), child_idx=child_idx, desc=desc, word_id=word_id,
                     weight=rng.random(n_nodes) * 5 + 0.01)


def bow_vectors(seed=SEED0 + 11, n_vec=64, n_words_vocab=100000, n_feat=2000, shared=0.4, group=8):
    """BowVectors as ComputeBoW leaves them (TF-IDF weights summed per word, L1-normalised in word order): vectors of one
    group (consecutive keyframes seeing the same place) draw `shared` of their words from a common pool."""
    from .views import BowVecData
    rng = np.random.default_rng(seed)
    idf = rng.random(n_words_vocab) * 8 + 0.05
    out = []
    pool = None
    for k in range(n_vec):
        if k % group == 0:
            pool = rng.integers(0, n_words_vocab, n_feat)
        words = np.where(rng.random(n_feat) < shared, pool[rng.integers(0, n_feat, n_feat)], rng.integers(0, n_words_vocab, n_feat))
        uniq, cnt = np.unique(words, return_counts=True)
        val = np.zeros(len(uniq))
        for i, (w, c) in enumerate(zip(uniq, cnt)):      # addWeight: one add per feature
            acc = 0.0
            for _ in range(int(c)):
                acc += idf[w]
            val[i] = acc
        norm = 0.0
        for v in val:
            norm += abs(v)
        out.append(BowVecData(word=uniq.astype(np.uint32), value=val / norm))
    return out
