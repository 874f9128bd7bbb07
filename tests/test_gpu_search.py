"""GPU parity of the search kernels through the C ABI against the CPU oracle (bit-exact):
a7/a8 SearchByBoW, a9 SearchForTriangulation, a6 SearchByProjection(F, MapPoints), a11 windowed best-1
overloads, a5 SearchForInitialization."""
import os

import numpy as np
import pytest

from orbslam2_amd import synth
from orbslam2_amd.views import MP_PRESENT, MP_OBS, MP_BAD, Q_VALID, Q_OBS, FrameData, ProjQueries

pytestmark = pytest.mark.gpu


def _fuzz_offset():
    """scripts/fuzz_more.py runs the fuzz tests below over many seeds by setting this."""
    return int(os.environ.get("ORBM_FUZZ_SEED", "0"))


def contended_pool(seed, n_kf, n, n_lm, **kw):
    """Keyframes whose keypoints re-observe a SMALL landmark pool: many keypoints of one keyframe share a
    landmark, so several queries want the same candidate and the in-order claim rules decide."""
    rng = np.random.default_rng(seed)
    lm_desc = synth.random_descriptors(rng, n_lm)
    frames = []
    for _ in range(n_kf):
        lm = rng.integers(0, n_lm, n)
        flags = np.where(rng.random(n) < kw.get("valid", 0.9), MP_PRESENT | MP_OBS, 0).astype(np.uint8)
        flags[(rng.random(n) < kw.get("bad", 0.05)) & (flags != 0)] |= MP_BAD
        ur = np.where(rng.random(n) < 0.3, rng.uniform(1, 600, n), -1.0).astype(np.float32)
        frames.append(synth.make_frame(rng, lm_desc, lm, mp_flags=flags, uright=ur, p_flip=kw.get("p_flip", 0.05),
                                       n_nodes=kw.get("n_nodes", 20), octave=rng.integers(0, 8, n).astype(np.int32)))
    return frames


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("check_ori", [True, False])
def test_bow_parity_pool(ctx, orc, variant, check_ori):
    kfs = synth.keyframe_pool(seed=synth.SEED0 + 4, n_kf=6, n=2000)
    ids = ctx.upload_frames(kfs)
    pa, pb = np.triu_indices(6, 1)
    counts, match = ctx.search_bow(variant, ids[pa], ids[pb], 0.75, check_ori, stride=2000)
    tot = 0
    for k, (i, j) in enumerate(zip(pa, pb)):
        wc, wm, _ = orc.search_bow(variant, kfs[i], kfs[j], 0.75, check_ori)
        assert counts[k] == wc, (i, j, counts[k], wc)
        assert np.array_equal(match[k], wm), (i, j)
        tot += wc
    assert tot > 1000
    ctx.free_frames(ids)


@pytest.mark.parametrize("variant", [0, 1])
def test_bow_large_batch_uses_one_workgroup_per_pair(ctx, orc, variant):
    """Large batches run one workgroup per pair, smaller ones split a pair's nodes over several workgroups (the library
    picks by batch size; the threshold is moved here to take both paths with the same 540 pairs): same results either way."""
    from orbslam2_amd import _lib
    L = _lib.load()
    kfs = synth.keyframe_pool(seed=synth.SEED0 + 14, n_kf=6, n=600)
    ids = ctx.upload_frames(kfs)
    pa, pb = np.meshgrid(np.arange(6), np.arange(6))
    pa, pb = np.resize(pa.ravel(), 540), np.resize(pb.ravel(), 540)
    want = {}
    try:
        for split_below in (512, 2048):
            L.orbm__bow_split_below(split_below)
            counts, match = ctx.search_bow(variant, ids[pa], ids[pb], 0.75, True, stride=600)
            for k, (i, j) in enumerate(zip(pa, pb)):
                if (i, j) not in want:
                    want[(i, j)] = orc.search_bow(variant, kfs[i], kfs[j], 0.75, True)
                wc, wm, _ = want[(i, j)]
                assert counts[k] == wc and np.array_equal(match[k], wm), (split_below, k, i, j)
    finally:
        L.orbm__bow_split_below(2048)
    ctx.free_frames(ids)


def test_triang_large_batch(ctx, orc):
    c = synth.triang_pair(77, n=800)
    ids = ctx.upload_frames(list(c[:2]))
    from orbslam2_amd import _lib
    L = _lib.load()
    P = 520
    wc, wm, _ = orc.search_triang(c[0], c[1], c[2], c[3], False, True)
    try:
        for split_below in (512, 2048):                      # one workgroup per pair / several
            L.orbm__bow_split_below(split_below)
            counts, match = ctx.search_triang(np.full(P, ids[0]), np.full(P, ids[1]), np.tile(c[2], (P, 1, 1)), np.tile(c[3], (P, 1)),
                                              False, True, stride=800)
            assert (counts == wc).all() and (match == wm[None, :]).all() and wc > 10
    finally:
        L.orbm__bow_split_below(2048)
    ctx.free_frames(ids)


@pytest.mark.parametrize("form", [1, 2, 5])
def test_bow_kernel_forms_are_all_exact(ctx, orc, form):
    """The BoW searches exist in three forms (candidates on lanes / queries on lanes / node blocks on the matrix cores; the
    library picks by batch size): all give the reference's result on every kind of input, for small and large batches."""
    from orbslam2_amd import _lib
    L = _lib.load()
    try:
        L.orbm__bow_variant(form)
        L.orbm__bow_split_below(512)                                      # 81 pairs: split path, 567 / 600: one workgroup per pair
        kfs = synth.keyframe_pool(seed=synth.SEED0 + 24, n_kf=4, n=1200) + contended_pool(31, 3, 700, 150) + \
            contended_pool(41, 2, 300, 10, n_nodes=2)                     # nodes far wider than a wavefront
        ids = ctx.upload_frames(kfs)
        pa, pb = np.meshgrid(np.arange(len(kfs)), np.arange(len(kfs)))
        pa, pb = pa.ravel(), pb.ravel()
        for reps in (1, 7):                                               # 81 pairs (split path) and 567 pairs
            qa, qb = np.tile(pa, reps), np.tile(pb, reps)
            for variant in (0, 1):
                counts, match = ctx.search_bow(variant, ids[qa], ids[qb], 0.8, True, stride=1200)
                for k in range(len(pa)):
                    i, j = pa[k], pb[k]
                    wc, wm, _ = orc.search_bow(variant, kfs[i], kfs[j], 0.8, True)
                    for r in range(reps):
                        kk = r * len(pa) + k
                        assert counts[kk] == wc and np.array_equal(match[kk, :len(wm)], wm), (form, reps, variant, i, j)
        ctx.free_frames(ids)
        cases = [synth.triang_pair(150 + s2, n=900) for s2 in range(2)]
        ids = ctx.upload_frames([f for c in cases for f in c[:2]])
        for reps in (1, 300):
            fa, fb = np.tile(ids[0::2], reps), np.tile(ids[1::2], reps)
            f12 = np.tile(np.stack([c[2] for c in cases]), (reps, 1, 1)); ep = np.tile(np.stack([c[3] for c in cases]), (reps, 1))
            for only_stereo in (False, True):
                counts, match = ctx.search_triang(fa, fb, f12, ep, only_stereo, True, stride=900)
                for k, c in enumerate(cases):
                    wc, wm, _ = orc.search_triang(c[0], c[1], c[2], c[3], only_stereo, True)
                    for r in range(reps):
                        assert counts[r * 2 + k] == wc and np.array_equal(match[r * 2 + k], wm), (form, reps, only_stereo, k)
        ctx.free_frames(ids)
    finally:
        L.orbm__bow_variant(0)
        L.orbm__bow_split_below(2048)


def test_bow_tile_list_passes(ctx, orc):
    """The matrix-core form lists a wave's 32 x 32 tiles in LDS, 128 at a time: frames whose nodes need many more (a single
    node of 1500 x 1500 features = 2209 tiles, 18 passes of one wave) and nodes of very unequal size side by side, so that a
    node's tiles straddle the end of a pass and the last entry of a pass is not the last tile of its node.  Both kinds,
    split and whole-pair launches, the rotation filter on."""
    from orbslam2_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(4242)
    lm_desc = synth.random_descriptors(rng, 400)
    def frame(n, node_sizes):
        lm = rng.integers(0, len(lm_desc), n)
        flags = np.where(rng.random(n) < 0.9, MP_PRESENT | MP_OBS, 0).astype(np.uint8)
        f = synth.make_frame(rng, lm_desc, lm, mp_flags=flags, p_flip=0.04, n_nodes=1, octave=rng.integers(0, 8, n).astype(np.int32))
        node = np.repeat(np.arange(len(node_sizes)) * 4, node_sizes)[:n]      # node ids 0, 4, 8, ...: every one lands in wave 0 .. 3 by its index, not its id
        node = np.concatenate([node, np.full(n - len(node), node[-1])]) if len(node) < n else node
        f.fv_node, f.fv_off, f.fv_idx = synth.feature_vector(rng.permutation(node))
        return f
    kfs = [frame(1500, [1500]), frame(1400, [1400]),                                   # one node each: 47 x 44 tiles in one wave
           frame(1500, [700, 33, 5, 600, 1, 161]), frame(1300, [650, 40, 7, 500, 3, 100]),   # 22 x 21 = 462 tiles, then 2 x 2, 1, 19 x 16, 1, 6 x 4
           frame(900, [31] * 29 + [1]), frame(950, [33] * 28 + [26])]                  # many small nodes: 1 tile against 2 x 2 per node
    ids = ctx.upload_frames(kfs)
    pa, pb = np.meshgrid(np.arange(len(kfs)), np.arange(len(kfs)))
    pa, pb = pa.ravel(), pb.ravel()
    try:
        L.orbm__bow_variant(5)
        for split_below in (1, 4096):                                  # whole pairs / pairs split over workgroups
            L.orbm__bow_split_below(split_below)
            for variant in (0, 1):
                counts, match = ctx.search_bow(variant, ids[pa], ids[pb], 0.75, True, stride=1500)
                for k in range(len(pa)):
                    wc, wm, _ = orc.search_bow(variant, kfs[pa[k]], kfs[pb[k]], 0.75, True)
                    assert counts[k] == wc and np.array_equal(match[k, :len(wm)], wm), (split_below, variant, pa[k], pb[k])
        assert sum(orc.search_bow(1, kfs[i], kfs[i + 1], 0.75, True)[0] for i in (0, 2, 4)) > 100
    finally:
        L.orbm__bow_variant(0)
        L.orbm__bow_split_below(2048)
    ctx.free_frames(ids)


def test_bow_largest_frames(ctx, orc):
    """Frames at the size limit (65535 keypoints: the largest row index a tile-list entry has to hold) against a large and a
    small one, nodes of ~1000 rows (tile lists of ~1000 entries per node, many passes), both kinds, the matrix-core form."""
    from orbslam2_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(65535)
    lm_desc = synth.random_descriptors(rng, 30000)
    def frame(n, n_nodes):
        lm = rng.integers(0, len(lm_desc), n)
        flags = np.where(rng.random(n) < 0.8, MP_PRESENT | MP_OBS, 0).astype(np.uint8)
        return synth.make_frame(rng, lm_desc, lm, mp_flags=flags, p_flip=0.05, n_nodes=n_nodes, node_noise=0.02, octave=rng.integers(0, 8, n).astype(np.int32))
    kfs = [frame(65535, 64), frame(40000, 64), frame(700, 64)]
    ids = ctx.upload_frames(kfs)
    try:
        L.orbm__bow_variant(5)
        for variant in (0, 1):
            pa = np.array([0, 1, 0, 2]); pb = np.array([1, 0, 2, 0])
            counts, match = ctx.search_bow(variant, ids[pa], ids[pb], 0.75, True, stride=65535)
            for k in range(len(pa)):
                wc, wm, _ = orc.search_bow(variant, kfs[pa[k]], kfs[pb[k]], 0.75, True)
                assert counts[k] == wc and np.array_equal(match[k, :len(wm)], wm), (variant, pa[k], pb[k])
                assert k > 1 or wc > 1000
    finally:
        L.orbm__bow_variant(0)
    ctx.free_frames(ids)


@pytest.mark.parametrize("ratio", [0.3, 0.6, 0.9, 1.0, 1.5])
def test_bow_matrix_core_form_ratios(ctx, orc, ratio):
    """The matrix-core form only looks at distances below a bound derived from the ratio (67 for 0.75, 84 for 0.6, 51 for
    ratios of 1 and more; above 128, as for 0.3, the library takes the vector-ALU form): every ratio gives the reference's
    result, on overlapping keyframes and on keyframes whose queries fight over candidates."""
    from orbslam2_amd import _lib
    L = _lib.load()
    kfs = synth.keyframe_pool(seed=synth.SEED0 + 34, n_kf=3, n=1500) + contended_pool(51, 3, 600, 120) + contended_pool(61, 2, 500, 400, p_flip=0.12)
    ids = ctx.upload_frames(kfs)
    pa, pb = np.meshgrid(np.arange(len(kfs)), np.arange(len(kfs)))
    pa, pb = pa.ravel(), pb.ravel()
    try:
        L.orbm__bow_variant(5)
        for variant in (0, 1):
            counts, match = ctx.search_bow(variant, ids[pa], ids[pb], ratio, True, stride=1500)
            tot = 0
            for k, (i, j) in enumerate(zip(pa, pb)):
                wc, wm, _ = orc.search_bow(variant, kfs[i], kfs[j], ratio, True)
                assert counts[k] == wc and np.array_equal(match[k, :len(wm)], wm), (ratio, variant, i, j, counts[k], wc)
                tot += wc
            assert tot > 500
    finally:
        L.orbm__bow_variant(0)
    ctx.free_frames(ids)


@pytest.mark.parametrize("variant", [0, 1])
def test_bow_parity_contended_claims(ctx, orc, variant):
    kfs = contended_pool(11 + variant, 5, 700, 150) + contended_pool(21, 2, 300, 10, n_nodes=2)  # nodes > 64 wide
    ids = ctx.upload_frames(kfs)
    pa, pb = np.meshgrid(np.arange(len(kfs)), np.arange(len(kfs)))
    pa, pb = pa.ravel(), pb.ravel()          # includes a keyframe against itself
    counts, match = ctx.search_bow(variant, ids[pa], ids[pb], 0.9, True, stride=700)
    for k, (i, j) in enumerate(zip(pa, pb)):
        wc, wm, _ = orc.search_bow(variant, kfs[i], kfs[j], 0.9, True)
        n_out = kfs[j].n if variant == 0 else kfs[i].n
        assert counts[k] == wc, (i, j, counts[k], wc)
        assert np.array_equal(match[k, :n_out], wm), (i, j)
    ctx.free_frames(ids)


def test_bow_empty_and_disjoint(ctx, orc):
    rng = np.random.default_rng(3)
    lm = synth.random_descriptors(rng, 100)
    fl = np.full(50, MP_PRESENT | MP_OBS, np.uint8)
    a = synth.make_frame(rng, lm, np.arange(50), mp_flags=fl)
    b = synth.make_frame(rng, lm, np.arange(50, 100), mp_flags=fl)   # other landmarks: nodes mostly differ
    e = synth.make_frame(rng, lm, np.arange(0), mp_flags=np.zeros(0, np.uint8))
    ids = ctx.upload_frames([a, b, e])
    for variant in (0, 1):
        for (i, j) in ((0, 1), (0, 2), (2, 0), (2, 2), (0, 0)):
            fr = [a, b, e]
            counts, match = ctx.search_bow(variant, [ids[i]], [ids[j]], 0.75, True, stride=50)
            wc, wm, _ = orc.search_bow(variant, fr[i], fr[j], 0.75, True)
            assert counts[0] == wc
            assert np.array_equal(match[0, :len(wm)], wm)
    ctx.free_frames(ids)


@pytest.mark.parametrize("only_stereo,check_ori", [(False, False), (False, True), (True, False)])
def test_triang_parity(ctx, orc, only_stereo, check_ori):
    cases = [synth.triang_pair(50 + s, n=1500) for s in range(3)]
    frames = [f for c in cases for f in c[:2]]
    ids = ctx.upload_frames(frames)
    fa, fb = ids[0::2], ids[1::2]
    f12 = np.stack([c[2] for c in cases]); ep = np.stack([c[3] for c in cases])
    counts, match = ctx.search_triang(fa, fb, f12, ep, only_stereo, check_ori, stride=1500)
    tot = 0
    for k, c in enumerate(cases):
        wc, wm, _ = orc.search_triang(c[0], c[1], c[2], c[3], only_stereo, check_ori)
        assert counts[k] == wc
        assert np.array_equal(match[k], wm)
        tot += wc
    assert tot > (10 if only_stereo else 200)
    ctx.free_frames(ids)


def test_triang_last_wins_on_equal_distance(ctx, orc):
    """dist > bestDist skips, so an equal distance REPLACES (ORBmatcher.cc:993): duplicate candidates."""
    f1, f2, f12, ep = synth.triang_pair(99, n=400, stereo_frac=1.0, has_mp_frac=0.0)
    # make every f2 keypoint appear twice (same descriptor, same position, same node)
    dup = FrameData(desc=np.concatenate([f2.desc, f2.desc]), x=np.concatenate([f2.x, f2.x]), y=np.concatenate([f2.y, f2.y]),
                    angle=np.concatenate([f2.angle, f2.angle]), octave=np.concatenate([f2.octave, f2.octave]),
                    uright=np.concatenate([f2.uright, f2.uright]), mp_flags=np.zeros(800, np.uint8))
    node = np.zeros(400, np.uint32)
    for k in range(len(f2.fv_node)):
        node[f2.fv_idx[f2.fv_off[k]:f2.fv_off[k + 1]]] = f2.fv_node[k]
    dup.fv_node, dup.fv_off, dup.fv_idx = synth.feature_vector(np.concatenate([node, node]))
    ids = ctx.upload_frames([f1, dup])
    counts, match = ctx.search_triang([ids[0]], [ids[1]], f12[None], ep[None], False, False, stride=400)
    wc, wm, _ = orc.search_triang(f1, dup, f12, ep, False, False)
    assert counts[0] == wc and np.array_equal(match[0], wm)
    assert (wm >= 400).sum() > 50          # the later duplicate is the one reported
    ctx.free_frames(ids)


@pytest.mark.parametrize("form", [1, 2, 5])
def test_triang_fuzz_crowded_nodes(ctx, orc, form):
    """SearchForTriangulation in all kernel forms on crowded nodes: small landmark pools (many candidates at or below
    TH_LOW for one query, equal distances: the last one wins), random stereo flags and map points, both only_stereo
    settings, small and large batches."""
    from orbslam2_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(777 + form + _fuzz_offset())
    kfs = []
    for trial in range(12):
        n = int(rng.integers(60, 500))
        lm_desc = synth.random_descriptors(rng, int(rng.integers(6, 60)))
        lm = rng.integers(0, len(lm_desc), n)
        flags = np.where(rng.random(n) < 0.4, MP_PRESENT | MP_OBS, 0).astype(np.uint8)
        ur = np.where(rng.random(n) < 0.4, rng.uniform(1, 600, n), -1.0).astype(np.float32)
        # a landmark sits on the same image row (+- a pixel) in every keyframe: with the horizontal epipolar lines below its
        # observations pass CheckDistEpipolarLine, at the coarse octaves also those of its neighbours
        y = (20.0 + (lm * 37 % 440) + rng.normal(0, 0.8, n)).astype(np.float32)
        kfs.append(synth.make_frame(rng, lm_desc, lm, y=y, mp_flags=flags, uright=ur, p_flip=float(rng.choice([0.03, 0.08, 0.1])),
                                    n_nodes=int(rng.choice([1, 4, 12])), octave=rng.integers(0, 8, n).astype(np.int32)))
    ids = ctx.upload_frames(kfs)
    P = 560
    pa = rng.integers(0, len(kfs), P); pb = rng.integers(0, len(kfs), P)
    f12 = np.tile(np.array([0, 0, 0, 0, 0, 1, 0, -1, 0], np.float32), (P, 1)) + rng.normal(0, 3e-6, (P, 9)).astype(np.float32)
    ep = np.stack([rng.uniform(0, 640, P), rng.uniform(0, 480, P)], axis=1).astype(np.float32)
    try:
        L.orbm__bow_variant(form)
        L.orbm__bow_split_below(512)                          # 560 pairs: one workgroup per pair, 30 pairs: split path
        for only_stereo in (False, True):
            want = [orc.search_triang(kfs[pa[k]], kfs[pb[k]], f12[k], ep[k], only_stereo, True) for k in range(P)]
            assert sum(w[0] for w in want) > 100
            for lo, hi in ((0, P), (0, 30)):
                counts, match = ctx.search_triang(ids[pa[lo:hi]], ids[pb[lo:hi]], f12[lo:hi].reshape(-1, 3, 3), ep[lo:hi], only_stereo, True, stride=500)
                for k in range(lo, hi):
                    wc, wm, _ = want[k]
                    assert counts[k - lo] == wc and np.array_equal(match[k - lo, :len(wm)], wm), (form, only_stereo, hi, k)
    finally:
        L.orbm__bow_variant(0)
        L.orbm__bow_split_below(2048)
    ctx.free_frames(ids)


@pytest.mark.parametrize("th", [1.0, 3.0, 5.0])
@pytest.mark.parametrize("stereo", [False, True])
def test_proj_points_parity(ctx, orc, th, stereo):
    f, q = synth.proj_instance(seed=300 + int(th) + 10 * stereo, n=2000, n_queries=6000, stereo=stereo,
                               claimed_frac=0.2, obs_frac=0.8, valid_frac=0.9)
    fid = ctx.upload_frames([f]); qid = ctx.upload_proj_queries([q])
    counts, owner = ctx.search_proj_points(fid, qid, th, 0.8, stride=2000)
    wc, wo, ncmp = orc.search_proj_points(f, q, th, 0.8)
    assert counts[0] == wc and wc > 500
    assert np.array_equal(owner[0], wo)
    ctx.free_frames(fid); ctx.free_queries(qid)


def test_proj_points_batch_with_slot_flags_and_edges(ctx, orc):
    insts = [synth.proj_instance(seed=400 + k, n=[2000, 1, 700, 64][k], n_queries=[5000, 10, 0, 3000][k],
                                 claimed_frac=0.1, obs_frac=0.5) for k in range(4)]
    # queries far outside the image and on the border cells
    insts[3][1].proj_x[:50] = np.linspace(-300, 900, 50, dtype=np.float32)
    insts[3][1].proj_y[:50] = np.linspace(900, -300, 50, dtype=np.float32)
    fids = ctx.upload_frames([i[0] for i in insts]); qids = ctx.upload_proj_queries([i[1] for i in insts])
    rng = np.random.default_rng(9)
    slot = np.zeros((4, 2000), np.uint8)
    slot[rng.random((4, 2000)) < 0.3] = MP_PRESENT | MP_OBS
    slot[rng.random((4, 2000)) < 0.1] = MP_PRESENT
    counts, owner = ctx.search_proj_points(fids, qids, 3.0, 0.8, stride=2000, slot_flags=slot)
    for k, (f, q) in enumerate(insts):
        wc, wo, _ = orc.search_proj_points(f, q, 3.0, 0.8, slot_flags=slot[k, :f.n])
        assert counts[k] == wc
        assert np.array_equal(owner[k, :f.n], wo)
        assert (owner[k, f.n:] == -1).all()
    ctx.free_frames(fids); ctx.free_queries(qids)


def test_proj_points_heavy_contention(ctx, orc):
    """Many queries per keypoint, unsorted, large windows: long claim chains inside a chunk."""
    f, q = synth.proj_instance(seed=555, n=300, n_queries=20000, th_noise=6.0, obs_frac=0.6)
    fid = ctx.upload_frames([f]); qid = ctx.upload_proj_queries([q])
    for th in (5.0, 1.0):
        counts, owner = ctx.search_proj_points(fid, qid, th, 0.9, stride=300)
        wc, wo, _ = orc.search_proj_points(f, q, th, 0.9)
        assert counts[0] == wc and np.array_equal(owner[0], wo)
    ctx.free_frames(fid); ctx.free_queries(qid)


@pytest.mark.parametrize("mode,dist_th", [(0, 100), (1, 100), (1, 64), (2, 50)])
@pytest.mark.parametrize("check_ori", [True, False])
def test_window_parity(ctx, orc, mode, dist_th, check_ori):
    insts = [synth.window_instance(700 + 10 * mode + k, mode, n=[2000, 500, 1][k], n_queries=[3000, 4000, 5][k],
                                   stereo=(k == 0), claimed_frac=0.15) for k in range(3)]
    fids = ctx.upload_frames([i[0] for i in insts]); qids = ctx.upload_win_queries([i[1] for i in insts])
    counts, owner = ctx.search_window(mode, fids, qids, dist_th, check_ori, stride=2000)
    tot = 0
    for k, (f, q) in enumerate(insts):
        wc, wo, _ = orc.search_window(mode, f, q, dist_th, check_ori)
        assert counts[k] == wc, (k, counts[k], wc)
        assert np.array_equal(owner[k, :f.n], wo)
        tot += wc
    assert tot > 300
    ctx.free_frames(fids); ctx.free_queries(qids)


@pytest.mark.parametrize("n,window,ratio,check_ori", [(1000, 100, 0.9, True), (1000, 10, 0.9, True),
                                                      (2000, 100, 0.9, False), (300, 400, 0.95, True)])
def test_search_init_parity(ctx, orc, n, window, ratio, check_ori):
    f1, f2, prev = synth.init_pair(seed=synth.SEED0 + 1 + n, n=n)
    ids = ctx.upload_frames([f1, f2])
    cnt, m12, prev_out = ctx.search_init(ids[0], ids[1], prev, window, ratio, check_ori)
    wc, wm, wprev, ncmp = orc.search_init(f1, f2, prev, window, ratio, check_ori)
    assert cnt == wc
    assert np.array_equal(m12, wm)
    assert np.array_equal(prev_out, wprev)
    if window >= 100:
        assert wc > 20
    ctx.free_frames(ids)


def test_search_init_steals(ctx, orc):
    """Several F1 keypoints see the same F2 keypoint: a later, closer one steals it (ORBmatcher.cc:629-633)."""
    rng = np.random.default_rng(8)
    lm = synth.random_descriptors(rng, 50)
    n = 400
    oct0 = np.zeros(n, np.int32)
    f2 = synth.make_frame(rng, lm, rng.integers(0, 50, n), with_fv=False, octave=oct0, p_flip=0.03,
                          x=rng.uniform(100, 200, n).astype(np.float32), y=rng.uniform(100, 200, n).astype(np.float32))
    f1 = synth.make_frame(rng, lm, rng.integers(0, 50, n), with_fv=False, octave=oct0, p_flip=0.03,
                          x=rng.uniform(100, 200, n).astype(np.float32), y=rng.uniform(100, 200, n).astype(np.float32))
    prev = np.stack([f1.x, f1.y], 1)
    ids = ctx.upload_frames([f1, f2])
    cnt, m12, prev_out = ctx.search_init(ids[0], ids[1], prev, 150, 1.5, True)
    wc, wm, wprev, _ = orc.search_init(f1, f2, prev, 150, 1.5, True)
    assert cnt == wc and np.array_equal(m12, wm) and np.array_equal(prev_out, wprev)
    ctx.free_frames(ids)


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 40])
def test_window_kernel_variants_are_all_exact(ctx, orc, variant):
    """The windowed searches have four forms (0 = chosen by batch shape; 3 = fused: one launch, a workgroup per instance searches
    and settles claims; 4 = split: collect + claims kernels -- with few queries in the call a query's candidates are spread
    over a wave's lanes, 40 = split without that; 1 = single kernel with the frame in LDS; 2 = single kernel reading
    global memory; frames above 2200 keypoints never use LDS staging): same results."""
    from orbslam2_amd import _lib
    L = _lib.load()
    try:
        L.orbm__window_variant(4 if variant == 40 else variant)
        L.orbm__window_coop(0 if variant == 40 else 1)
        for seed, n, nq, th, stereo in ((901, 3000, 8000, 3.0, False), (902, 800, 5000, 5.0, True), (904, 2000, 3000, 1.0, False),
                                         (905, 2000, 3000, 8.0, True)):   # th 8: more candidates than a row holds
            f, q = synth.proj_instance(seed=seed, n=n, n_queries=nq, claimed_frac=0.1, obs_frac=0.7, stereo=stereo)
            fid = ctx.upload_frames([f]); qid = ctx.upload_proj_queries([q])
            counts, owner = ctx.search_proj_points(fid, qid, th, 0.8, stride=n)
            wc, wo, _ = orc.search_proj_points(f, q, th, 0.8)
            assert counts[0] == wc and np.array_equal(owner[0], wo), (seed, variant)
            ctx.free_frames(fid); ctx.free_queries(qid)
        for mode in (0, 1, 2):
            for wth in (7.0, 15.0):
                f, q = synth.window_instance(960 + mode, mode, n=1500, n_queries=2500, stereo=(mode == 0), th=wth)
                fid = ctx.upload_frames([f]); qid = ctx.upload_win_queries([q])
                counts, owner = ctx.search_window(mode, fid, qid, 100 if mode < 2 else 50, mode < 2, stride=1500)
                wc, wo, _ = orc.search_window(mode, f, q, 100 if mode < 2 else 50, mode < 2)
                assert counts[0] == wc and np.array_equal(owner[0], wo), (mode, wth, variant)
                ctx.free_frames(fid); ctx.free_queries(qid)
    finally:
        L.orbm__window_variant(0)
        L.orbm__window_coop(1)


@pytest.mark.parametrize("n", [2040, 2100, 2124, 2160, 3300, 4900, 5300, 6500, 8100])
def test_proj_points_across_the_lds_layout_boundaries(ctx, orc, n):
    """collect_kernel keeps the frame's descriptors in LDS up to ~2100 keypoints, reads them from global memory above that,
    and hands over to the single-kernel forms when its index no longer fits: frame sizes on both sides of each boundary
    (th 1 and th 5 size the bucket structures differently), alone and mixed with a small frame in the same launch."""
    f, q = synth.proj_instance(seed=7000 + n, n=n, n_queries=4000, claimed_frac=0.1, obs_frac=0.7, stereo=(n % 2 == 0))
    f2, q2 = synth.proj_instance(seed=7001 + n, n=300, n_queries=900, claimed_frac=0.1, obs_frac=0.7)
    fid = ctx.upload_frames([f, f2]); qid = ctx.upload_proj_queries([q, q2])
    for th in (1.0, 5.0):
        counts, owner = ctx.search_proj_points(fid, qid, th, 0.8, stride=n)
        for k, (ff, qq) in enumerate(((f, q), (f2, q2))):
            wc, wo, _ = orc.search_proj_points(ff, qq, th, 0.8)
            assert counts[k] == wc and np.array_equal(owner[k, :ff.n], wo), (n, th, k)
    ctx.free_frames(fid); ctx.free_queries(qid)


def test_frames_over_the_window_limit_are_rejected(ctx):
    from orbslam2_amd.matcher import OrbmError
    f, q = synth.proj_instance(seed=903, n=9000, n_queries=10)
    fid = ctx.upload_frames([f]); qid = ctx.upload_proj_queries([q])
    with pytest.raises(OrbmError):
        ctx.search_proj_points(fid, qid, 3.0, 0.8, stride=9000)
    ctx.free_frames(fid); ctx.free_queries(qid)


@pytest.mark.parametrize("chi2,dist_th,stereo", [(False, 100, False), (True, 50, True), (True, 50, False), (False, 50, True)])
def test_window_free_parity(ctx, orc, chi2, dist_th, stereo):
    """a12: claim-free best-1 (SearchBySim3 / Fuse inner loops), with and without the chi-square gate."""
    insts = []
    for k in range(3):
        f, q = synth.window_instance(950 + k + 10 * chi2, 2, n=[2000, 3000, 40][k], n_queries=[2500, 700, 300][k],
                                     stereo=stereo, th=[10.0, 3.0, 10.0][k])
        rng = np.random.default_rng(k)
        src_ur = np.where(rng.random(q.n) < 0.8, rng.uniform(0, 600, q.n), -1.0).astype(np.float32)
        q.ur = src_ur                      # u - bf*invz of the query (Fuse's stereo residual)
        insts.append((f, q))
    fids = ctx.upload_frames([i[0] for i in insts]); qids = ctx.upload_win_queries([i[1] for i in insts])
    best = ctx.search_window_free(fids, qids, dist_th, chi2, stride_q=2500)
    tot = 0
    for k, (f, q) in enumerate(insts):
        wb, _ = orc.search_window_free(f, q, dist_th, chi2)
        assert np.array_equal(best[k, :q.n], wb), k
        assert (best[k, q.n:] == -1).all()
        tot += (wb >= 0).sum()
    assert tot > 200
    # the same sets travelling with the call (orbm_search_window_free_once), and one set alone (no ragged rows)
    best1 = ctx.search_window_free_once(fids, [i[1] for i in insts], dist_th, chi2, stride_q=2500)
    assert np.array_equal(best1, best)
    for k, (f, q) in enumerate(insts):
        b = ctx.search_window_free_once(fids[k:k + 1], [q], dist_th, chi2, stride_q=q.n)
        assert np.array_equal(b[0], best[k, :q.n]), k
    ctx.free_frames(fids); ctx.free_queries(qids)


def test_distinctive_descriptors_parity(ctx, orc):
    """MapPoint::ComputeDistinctiveDescriptors batched over map points: empty lists, single observations, ties
    (identical descriptors), lists above one wavefront, and on both sides of 512 observations, where the kernel changes
    from its register form to the streaming form (the reference has no limit)."""
    rng = np.random.default_rng(77)
    sizes = [0, 1, 2, 3, 5, 8, 64, 65, 130, 512, 513, 700, 1100] + list(rng.integers(1, 40, 400))
    descs, off = [], [0]
    for n in sizes:
        lm = synth.random_descriptors(rng, 1)[0]
        d = np.stack([synth.observe(rng, lm[None, :], 0.08)[0] for _ in range(n)]) if n else np.zeros((0, 32), np.uint8)
        if n >= 3 and rng.random() < 0.3:
            d[rng.integers(0, n)] = d[0]          # exact duplicates: equal medians, first wins
        descs.append(d); off.append(off[-1] + n)
    got = ctx.distinctive_descriptors(np.array(off), np.concatenate(descs))
    want = np.array([orc.distinctive_descriptor(d) for d in descs])
    assert np.array_equal(got, want)
    assert (want[13:] >= 0).all() and len(set(want[13:])) > 5
    # all-equal descriptors above the limit: every median is 0, the first row wins
    assert ctx.distinctive_descriptors(np.array([0, 513]), np.zeros((513, 32), np.uint8))[0] == 0


def test_stereo_match_parity(ctx, orc):
    """Descriptor stage of Frame::ComputeStereoMatches: rectified stereo pair (same rows +- a pixel or two, disparity
    0..60), octaves +-1, some left keypoints with negative x, empty right frame."""
    rng = np.random.default_rng(88)
    pairs = []
    for n in (1500, 300, 64):
        lm = synth.random_descriptors(rng, n)
        xl = rng.uniform(16, 624, n).astype(np.float32); yl = rng.uniform(16, 464, n).astype(np.float32)
        octl = synth.octaves_level_major(n)
        left = synth.make_frame(rng, lm, np.arange(n), x=xl, y=yl, octave=octl, with_fv=False)
        keep = rng.random(n) < 0.8                       # 80 % of the landmarks are seen by the right camera
        m = int(keep.sum())
        disp = rng.uniform(-2, 60, m).astype(np.float32)
        xr = (xl[keep] - disp).astype(np.float32); yr = (yl[keep] + rng.normal(0, 1.2, m)).astype(np.float32)
        octr = np.clip(octl[keep] + rng.integers(-2, 3, m), 0, 7).astype(octl.dtype)
        extra = n // 3                                   # unrelated right keypoints
        lm2 = synth.random_descriptors(rng, extra)
        right_lm = np.concatenate([lm[keep], lm2])
        right = synth.make_frame(rng, right_lm, np.arange(m + extra),
                                 x=np.concatenate([xr, rng.uniform(16, 624, extra).astype(np.float32)]),
                                 y=np.concatenate([yr, rng.uniform(16, 464, extra).astype(np.float32)]),
                                 octave=np.concatenate([octr, synth.octaves_level_major(extra)]), with_fv=False)
        perm = rng.permutation(right.n)                  # right keypoints in extractor order, not left order
        right = synth.make_frame(rng, right.desc[perm], np.arange(right.n), x=right.x[perm], y=right.y[perm], octave=right.octave[perm],
                                 p_flip=0.0, with_fv=False)
        left.x[:3] = -1.0                                # maxU < 0 (:693)
        pairs.append((left, right))
    empty = synth.make_frame(rng, synth.random_descriptors(rng, 1), np.arange(0), with_fv=False)
    pairs.append((pairs[1][0], empty))
    ids = ctx.upload_frames([f for pr in pairs for f in pr])
    md = np.array([40.0 / 0.6, 35.0, 50.0, 35.0], np.float32)
    idx, dist = ctx.stereo_match(ids[0::2], ids[1::2], md, stride=1500)
    nmatch = 0
    for k, (l, r) in enumerate(pairs):
        wi, wd, _ = orc.stereo_match(l, r, 480, float(md[k]))
        assert np.array_equal(idx[k, :l.n], wi), k
        assert np.array_equal(dist[k, :l.n], wd), k
        nmatch += int((wd < 75).sum())
    assert nmatch > 800
    # a batch large enough for the four-wave form of the kernels (the small batch above runs the one-wave form), and
    # the scan form (used for right frames above 4096 keypoints) next to the row-binned one
    from orbslam2_amd import _lib
    rep = 200
    for scan_form in (0, 1):
        _lib.load().orbm__stereo_variant(scan_form)
        try:
            i2, d2 = ctx.stereo_match(ids[0::2], ids[1::2], md, stride=1500)
            big_i, big_d = ctx.stereo_match(np.tile(ids[0::2], rep), np.tile(ids[1::2], rep), np.tile(md, rep), stride=1500)
        finally:
            _lib.load().orbm__stereo_variant(0)
        assert np.array_equal(i2, idx) and np.array_equal(d2, dist), scan_form
        assert np.array_equal(big_i, np.tile(idx, (rep, 1))) and np.array_equal(big_d, np.tile(dist, (rep, 1))), scan_form
    ctx.free_frames(ids)
    # rows at the edges of the row bins (bands reaching below row 0, rows beyond the 2048 bins) and a right frame above
    # the 4096 keypoints the in-LDS sort holds (scan form chosen by the library)
    edge = []
    for n, ylo, yhi in ((400, 0.0, 9.0), (400, 2030.0, 2075.0), (5000, 16.0, 464.0)):
        lm = synth.random_descriptors(rng, n)
        xl = rng.uniform(16, 624, n).astype(np.float32); yl = rng.uniform(ylo, yhi, n).astype(np.float32)
        l = synth.make_frame(rng, lm, np.arange(n), x=xl, y=yl, with_fv=False, bounds=(0.0, 0.0, 640.0, 2100.0))
        perm = rng.permutation(n)
        r = synth.make_frame(rng, lm, perm, x=(xl - rng.uniform(0, 40, n).astype(np.float32))[perm],
                             y=(yl + rng.normal(0, 1.5, n).astype(np.float32))[perm], octave=l.octave[perm], with_fv=False,
                             bounds=(0.0, 0.0, 640.0, 2100.0))
        edge.append((l, r))
    ids = ctx.upload_frames([f for pr in edge for f in pr])
    idx, dist = ctx.stereo_match(ids[0::2], ids[1::2], np.full(3, 45.0, np.float32), stride=5000)
    for k, (l, r) in enumerate(edge):
        wi, wd, _ = orc.stereo_match(l, r, 4096, 45.0)
        assert np.array_equal(idx[k, :l.n], wi) and np.array_equal(dist[k, :l.n], wd), k
        assert (idx[k, l.n:] == -1).all() and (dist[k, l.n:] == 100).all() and (wi >= 0).sum() > l.n // 2
    ctx.free_frames(ids)


def _bits(idx):
    d = np.zeros(32, np.uint8)
    for i in idx:
        d[i >> 3] |= 1 << (i & 7)
    return d


def test_proj_second_best_above_threshold_is_not_claimed_early(ctx, orc):
    """Adversarial order dependence (SearchByProjection(F, MapPoints), ratio test on the second-best):
    keypoints a, b, c; queries i < j < k.  i takes a but could also take b (so j waits for i); k's best is c, which
    nobody lower could take (j's distance to c is 105 > TH_HIGH).  j's outcome still depends on c: with c free its
    second-best is 105 and the ratio test rejects it (90 > 0.8 * 105), with c taken it would be accepted.  A later
    query must therefore not take c before j is settled."""
    Sb, Sc, rest = list(range(0, 70)), list(range(70, 141)), list(range(141, 256))
    a, b, c = _bits([]), _bits(Sb), _bits(Sc)
    e = _bits(Sc[:45] + rest[60:100])               # k's second-best (46); out of reach of i and j (115, 119)
    qi = _bits(Sb[:30])                              # d(a) = 30, d(b) = 40, d(c) = 101
    qj = _bits(Sb[:40] + Sc[:33] + rest[:27])        # d(b) = 90, d(c) = 105, d(a) = 100
    qk = _bits(Sc[:51])                              # d(c) = 20, d(a) = 51, d(b) = 121
    assert [orc.hamming256(qi, v) for v in (a, b, c)] == [30, 40, 101]
    assert [orc.hamming256(qj, v) for v in (a, b, c)] == [100, 90, 105]
    assert orc.hamming256(qk, c) == 20 and orc.hamming256(qk, e) == 46 and orc.hamming256(qk, a) == 51
    assert orc.hamming256(qi, e) == 115 and orc.hamming256(qj, e) == 119
    # padding queries in front so that the three land in one chunk at arbitrary lanes, and far-away keypoints
    rng = np.random.default_rng(5)
    n_far = 40
    f = FrameData(desc=np.concatenate([np.stack([a, b, c, e]), synth.random_descriptors(rng, n_far)]),
                  x=np.concatenate([[300.0, 303.0, 306.0, 301.0], rng.uniform(20, 100, n_far)]).astype(np.float32),
                  y=np.concatenate([[200.0, 202.0, 204.0, 205.0], rng.uniform(20, 100, n_far)]).astype(np.float32),
                  angle=np.zeros(4 + n_far, np.float32), octave=np.full(4 + n_far, 2, np.int32))
    for pad in (0, 5, 70, 1030):
        nq = pad + 3
        desc = np.concatenate([synth.random_descriptors(rng, pad), np.stack([qi, qj, qk])]) if pad else np.stack([qi, qj, qk])
        px = np.concatenate([rng.uniform(400, 600, pad), [302.0, 303.0, 304.0]]).astype(np.float32)
        py = np.concatenate([rng.uniform(300, 450, pad), [201.0, 202.0, 203.0]]).astype(np.float32)
        q = ProjQueries(desc=desc, proj_x=px, proj_y=py, proj_xr=np.full(nq, -1.0, np.float32), level=np.full(nq, 2, np.int32),
                        view_cos=np.full(nq, 0.95, np.float32), flags=np.full(nq, Q_VALID | Q_OBS, np.uint8))
        wc, wo, _ = orc.search_proj_points(f, q, 3.0, 0.8)
        assert wo[0] == pad and wo[1] == -1 and wo[2] == pad + 2 and wo[3] == -1 and wc == 2      # a <- i, b free, c <- k
        fid = ctx.upload_frames([f]); qid = ctx.upload_proj_queries([q])
        for variant in (0, 1, 2, 3, 4):
            from orbslam2_amd import _lib
            _lib.load().orbm__window_variant(variant)
            try:
                counts, owner = ctx.search_proj_points(fid, qid, 3.0, 0.8, stride=f.n)
            finally:
                _lib.load().orbm__window_variant(0)
            assert counts[0] == wc and np.array_equal(owner[0], wo), (pad, variant, owner[0][:4])
        ctx.free_frames(fid); ctx.free_queries(qid)


def _dense_frame(rng, n, lm_desc):
    lm = rng.integers(0, len(lm_desc), n)
    return synth.make_frame(rng, lm_desc, lm, x=rng.uniform(200, 300, n).astype(np.float32), y=rng.uniform(200, 300, n).astype(np.float32),
                            octave=rng.integers(2, 4, n).astype(np.int32), p_flip=0.02, with_fv=False,
                            mp_flags=np.where(rng.random(n) < 0.1, MP_PRESENT | MP_OBS, 0).astype(np.uint8)), lm


def _claim_rounds_fuzz(ctx, orc):
    """(fused and split form of the windowed searches.)  Many small, crowded instances whose query-to-keypoint distances straddle the acceptance thresholds and whose
    windows overlap almost completely: the in-order claim rules decide nearly every match (a6 with its ratio test on
    the second-best, a11 best-only, a5 with stealing)."""
    rng = np.random.default_rng(2024 + _fuzz_offset())
    frames, pqs, wqs, inits = [], [], [], []
    for trial in range(160):
        n = int(rng.integers(20, 90)); nq = int(rng.integers(50, 400))
        lm_desc = synth.random_descriptors(rng, int(rng.integers(5, 40)))
        f, lm = _dense_frame(rng, n, lm_desc)
        frames.append(f)
        src = rng.integers(0, n, nq)
        p_flip = rng.choice([0.05, 0.2, 0.3, 0.36, 0.4, 0.44], nq)
        desc = np.stack([synth.observe(rng, lm_desc[lm[s]][None, :], p)[0] for s, p in zip(src, p_flip)])
        px = (f.x[src] + rng.normal(0, 6, nq)).astype(np.float32); py = (f.y[src] + rng.normal(0, 6, nq)).astype(np.float32)
        pqs.append(ProjQueries(desc=desc, proj_x=px, proj_y=py, proj_xr=np.full(nq, -1.0, np.float32),
                               level=np.clip(f.octave[src] + rng.integers(0, 2, nq), 0, 7).astype(np.int32),
                               view_cos=rng.uniform(0.9, 1.0, nq).astype(np.float32),
                               flags=np.where(rng.random(nq) < 0.95, Q_VALID, 0).astype(np.uint8) | np.where(rng.random(nq) < 0.7, Q_OBS, 0).astype(np.uint8)))
        from orbslam2_amd.views import WinQueries
        wqs.append(WinQueries(desc=desc, u=px, v=py, radius=rng.uniform(15, 60, nq).astype(np.float32),
                              min_level=(f.octave[src] - 1).astype(np.int32), max_level=(f.octave[src] + 1).astype(np.int32),
                              ur=None, angle=rng.uniform(0, 360, nq).astype(np.float32),
                              flags=np.where(rng.random(nq) < 0.95, Q_VALID | Q_OBS, 0).astype(np.uint8)))
    fid = ctx.upload_frames(frames)
    qid = ctx.upload_proj_queries(pqs)
    for th in (3.0, 6.0):
        counts, owner = ctx.search_proj_points(fid, qid, th, 0.8, stride=90)
        for k, (f, q) in enumerate(zip(frames, pqs)):
            wc, wo, _ = orc.search_proj_points(f, q, th, 0.8)
            assert counts[k] == wc and np.array_equal(owner[k, :f.n], wo), ("proj", th, k)
    ctx.free_queries(qid)
    wid = ctx.upload_win_queries(wqs)
    for mode, dth in ((0, 100), (1, 90), (2, 50), (2, 110)):
        counts, owner = ctx.search_window(mode, fid, wid, dth, mode < 2, stride=90)
        for k, (f, q) in enumerate(zip(frames, wqs)):
            wc, wo, _ = orc.search_window(mode, f, q, dth, mode < 2)
            assert counts[k] == wc and np.array_equal(owner[k, :f.n], wo), ("win", mode, dth, k)
    ctx.free_queries(wid); ctx.free_frames(fid)
    # a5: level-0 keypoints only, window covering everything, distances around TH_LOW, ratio near 1
    for trial in range(40):
        n1 = int(rng.integers(30, 200)); n2 = int(rng.integers(20, 120))
        lm_desc = synth.random_descriptors(rng, int(rng.integers(5, 30)))
        l1 = rng.integers(0, len(lm_desc), n1); l2 = rng.integers(0, len(lm_desc), n2)
        mk = lambda lm, n: synth.make_frame(rng, lm_desc, lm, x=rng.uniform(250, 330, n).astype(np.float32),
                                            y=rng.uniform(200, 280, n).astype(np.float32), octave=np.where(rng.random(n) < 0.8, 0, 1).astype(np.int32),
                                            p_flip=float(rng.choice([0.05, 0.1, 0.12])), with_fv=False)
        f1, f2 = mk(l1, n1), mk(l2, n2)
        prev = np.stack([f1.x, f1.y], 1).astype(np.float32)
        ids = ctx.upload_frames([f1, f2])
        for ratio in (0.9, 0.99):
            cnt, m12, prev_out = ctx.search_init(ids[0], ids[1], prev, 100, ratio, True)
            wc, wm, wprev, _ = orc.search_init(f1, f2, prev, 100, ratio, True)
            assert cnt == wc and np.array_equal(m12, wm) and np.array_equal(prev_out, wprev), ("init", trial, ratio)
        ctx.free_frames(ids)


@pytest.mark.parametrize("win_variant", [3, 4])
def test_claim_rounds_fuzz_near_thresholds(ctx, orc, win_variant):
    from orbslam2_amd import _lib
    _lib.load().orbm__window_variant(win_variant)
    try:
        _claim_rounds_fuzz(ctx, orc)
    finally:
        _lib.load().orbm__window_variant(0)


@pytest.mark.parametrize("form", [1, 2, 5])
def test_bow_fuzz_near_thresholds(ctx, orc, form):
    """All BoW kernel forms on crowded nodes whose distances straddle TH_LOW and the ratio test."""
    from orbslam2_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(4242 + form + _fuzz_offset())
    kfs = []
    for trial in range(24):
        n = int(rng.integers(30, 400))
        lm_desc = synth.random_descriptors(rng, int(rng.integers(4, 40)))
        lm = rng.integers(0, len(lm_desc), n)
        flags = np.where(rng.random(n) < 0.9, MP_PRESENT | MP_OBS, 0).astype(np.uint8)
        flags[(rng.random(n) < 0.05) & (flags != 0)] |= MP_BAD
        kfs.append(synth.make_frame(rng, lm_desc, lm, mp_flags=flags, p_flip=float(rng.choice([0.05, 0.1, 0.12])),
                                    n_nodes=int(rng.choice([1, 3, 10])), octave=rng.integers(0, 8, n).astype(np.int32)))
    ids = ctx.upload_frames(kfs)
    pa = rng.integers(0, len(kfs), 600); pb = rng.integers(0, len(kfs), 600)
    try:
        L.orbm__bow_variant(form)
        L.orbm__bow_split_below(512)                          # 600 pairs: one workgroup per pair, 40 pairs: split path
        for variant in (0, 1):
            for ratio in (0.75, 0.95):
                counts, match = ctx.search_bow(variant, ids[pa], ids[pb], ratio, True, stride=400)
                want = {}
                for k, (i, j) in enumerate(zip(pa, pb)):
                    if (i, j) not in want:
                        want[(i, j)] = orc.search_bow(variant, kfs[i], kfs[j], ratio, True)
                    wc, wm, _ = want[(i, j)]
                    assert counts[k] == wc and np.array_equal(match[k, :len(wm)], wm), (form, variant, ratio, k)
                counts, match = ctx.search_bow(variant, ids[pa[:40]], ids[pb[:40]], ratio, True, stride=400)   # split path
                for k in range(40):
                    wc, wm, _ = want[(pa[k], pb[k])]
                    assert counts[k] == wc and np.array_equal(match[k, :len(wm)], wm), (form, variant, ratio, k, "split")
    finally:
        L.orbm__bow_variant(0)
        L.orbm__bow_split_below(2048)
    ctx.free_frames(ids)


def test_all_pairs_end_to_end_on_one_gpu(ctx, orc):
    """BASELINE config 5 in small: every keyframe pair of a resident store through allpairs.all_pairs (device index lists,
    compact records left on the device by orbm_allpairs_batch), one rank; every record against the oracle."""
    import torch
    from orbslam2_amd import allpairs
    n_kf, n, threshold = 20, 500, 15
    kfs = synth.keyframe_pool(seed=synth.SEED0 + 5, n_kf=n_kf, n=n, window=int(n / 0.6))
    ids = ctx.upload_frames(kfs)
    dev = torch.device("cuda:0")
    tid = torch.tensor(ids, device=dev, dtype=torch.int32)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        fn = allpairs.device_batch_fn(torch, ctx, tid, 1, 0.75, 50, True, n, threshold, list_cap=64 * n, comm_device=dev)
        res = allpairs.all_pairs(torch, None, n_kf, batch=64, threshold=threshold, batch_fn=fn, tile=8, comm_device=dev)
    finally:
        ctx.use_own_stream()
    assert len(res) == n_kf * (n_kf - 1) // 2
    counts = res.counts()
    tot = n_listed = 0
    for (a, b), cnt in counts.items():
        wc, wm, _ = orc.search_bow(1, kfs[a], kfs[b], 0.75, True)
        assert cnt == wc, (a, b)
        row = res.match_row(a, b, n)
        if wc >= threshold:
            assert np.array_equal(row, wm), (a, b)
            n_listed += 1
        else:
            assert row is None
        tot += wc
    assert tot > 2000 and n_listed > 10
    ctx.free_frames(ids)


def _vocab_cases():
    return [dict(k=10, depth=3), dict(k=10, depth=4, flip=0.12), dict(k=6, depth=5, irregular=True, tie_frac=0.2),
            dict(k=3, depth=6, irregular=True, tie_frac=0.3, stop_frac=0.2), dict(k=2, depth=1)]


@pytest.mark.gpu
@pytest.mark.parametrize("form", [1, 4])
@pytest.mark.parametrize("case", range(5))
def test_vocabulary_transform_parity(ctx, orc, case, form):
    """DBoW2 transform on the device (Frame::ComputeBoW / KeyFrame::ComputeBoW): word ids, node ids and the FeatureVector
    left in the frame, against the restated DBoW2 descent; regular and irregular trees, exact ties between siblings,
    stopped words, levelsup from the leaves to above the root, empty frames, frames of different sizes in one call; both
    kernel forms (1 and 4 descents per lane group; the library picks by size)."""
    from orbslam2_amd import _lib
    kw = _vocab_cases()[case]
    vocab = synth.make_vocabulary(seed=500 + case, **kw)
    # DBoW2 weighting / scoring of the BowVector: TF_IDF + L1 (ORBvoc.txt), TF + DOT_PRODUCT (divide by the word count),
    # IDF + L1 and BINARY + DOT_PRODUCT (first weight only), TF_IDF + CHI_SQUARE (L1 norm)
    vocab.weighting, vocab.scoring = [(0, 0), (1, 5), (2, 0), (3, 5), (0, 2)][case]
    rng = np.random.default_rng(900 + case)
    sizes = [2000, 1, 0, 257, 1023, 4096 if case == 0 else 700]
    frames = []
    for n in sizes:
        d = synth.vocabulary_features(rng, vocab, n) if n else np.zeros((0, 32), np.uint8)
        frames.append(synth.make_frame(rng, d, np.arange(n), p_flip=0.0, with_fv=(n == 257)))   # one arrives with a host FV
    ids = ctx.upload_frames(frames)
    vid = ctx.upload_vocabulary(vocab)
    _lib.load().orbm__vocab_variant(form)
    try:
        for levelsup in ([4, 0, 1, 2, vocab.depth, vocab.depth + 3] if case < 3 else [4, 1]):
            word, node, bvids = ctx.compute_bow(vid, ids, levelsup, stride=max(sizes), want_bowvecs=True)
            for k, f in enumerate(frames):
                w_want, n_want, (fn, fo, fi), (bw, bv) = orc.transform(vocab, f.desc, levelsup, want_bow=True)
                gw, gv = ctx.download_bowvec(bvids[k])
                assert np.array_equal(gw, bw) and np.array_equal(gv.view(np.uint64), bv.view(np.uint64)), (case, levelsup, k)
                assert np.array_equal(word[k, :f.n], w_want), (case, levelsup, k)
                assert np.array_equal(node[k, :f.n], n_want), (case, levelsup, k)
                assert (word[k, f.n:] == -1).all() and (node[k, f.n:] == -1).all()
                gn, go, gi = ctx.feature_vector(ids[k], f.n)
                assert np.array_equal(gn, fn) and np.array_equal(go, fo) and np.array_equal(gi, fi), (case, levelsup, k)
                if k < 2 or f.n == 0:                      # the same descent on bare descriptors
                    w2, n2 = ctx.transform(vid, f.desc, levelsup)
                    assert np.array_equal(w2, w_want) and np.array_equal(n2, n_want)
            ctx.free_bowvecs(bvids)
        # the same call without the id outputs leaves the same FeatureVectors
        ctx.compute_bow(vid, ids[:2], 2, want_ids=False)
        _, _, (fn, fo, fi) = orc.transform(vocab, frames[0].desc, 2)
        gn, go, gi = ctx.feature_vector(ids[0], frames[0].n)
        assert np.array_equal(gn, fn) and np.array_equal(go, fo) and np.array_equal(gi, fi)
    finally:
        _lib.load().orbm__vocab_variant(0)
        ctx.free_vocabulary(vid)
        ctx.free_frames(ids)


@pytest.mark.gpu
def test_search_bow_on_device_built_feature_vectors(ctx, orc):
    """Keyframes uploaded WITHOUT a FeatureVector, ComputeBoW on the device, then SearchByBoW: the matches equal the
    oracle's on frames that carry the FeatureVector the restated DBoW2 builds on the host."""
    vocab = synth.make_vocabulary(seed=61, k=10, depth=5, flip=0.15)
    rng = np.random.default_rng(62)
    lm = synth.vocabulary_features(rng, vocab, 3000, flip=0.05)
    frames = []
    for _ in range(6):
        sel = rng.permutation(3000)[:1500]
        flags = np.where(rng.random(1500) < 0.9, MP_PRESENT, 0).astype(np.uint8)
        frames.append(synth.make_frame(rng, lm, sel, with_fv=False, mp_flags=flags))
    ids = ctx.upload_frames(frames)
    vid = ctx.upload_vocabulary(vocab)
    try:
        ctx.compute_bow(vid, ids, 4, want_ids=False)
        for f in frames:
            _, _, (f.fv_node, f.fv_off, f.fv_idx) = orc.transform(vocab, f.desc, 4)
        pa = np.array([0, 0, 1, 2, 3, 4], np.int32); pb = np.array([1, 2, 3, 4, 5, 5], np.int32)
        for variant in (0, 1):
            counts, match = ctx.search_bow(variant, ids[pa], ids[pb], 0.75, True, stride=1500)
            for p in range(len(pa)):
                cnt, m, _ = orc.search_bow(variant, frames[pa[p]], frames[pb[p]], 0.75, True)
                assert counts[p] == cnt and np.array_equal(match[p, :frames[pa[p]].n], m), (variant, p)
            assert counts.sum() > 100
    finally:
        ctx.free_vocabulary(vid)
        ctx.free_frames(ids)


@pytest.mark.gpu
def test_vocabulary_errors(ctx):
    from orbslam2_amd.matcher import OrbmError
    from orbslam2_amd.views import VocabData
    good = synth.make_vocabulary(seed=5, k=3, depth=2)
    bad_child = VocabData(depth=good.depth, child_off=good.child_off, child_idx=good.child_idx.copy(), desc=good.desc, word_id=good.word_id)
    bad_child.child_idx[2] = 0                                  # a child that is not a later node: the descent could loop
    with pytest.raises(OrbmError):
        ctx.upload_vocabulary(bad_child)
    no_word = VocabData(depth=good.depth, child_off=good.child_off, child_idx=good.child_idx, desc=good.desc, word_id=np.full(good.n_nodes, -1))
    with pytest.raises(OrbmError):
        ctx.upload_vocabulary(no_word)
    rng = np.random.default_rng(1)
    ids = ctx.upload_frames([synth.make_frame(rng, synth.random_descriptors(rng, 10), np.arange(10), with_fv=False)])
    vid = ctx.upload_vocabulary(good)
    with pytest.raises(OrbmError):
        ctx.compute_bow(vid + 7, ids, 4, stride=10)
    with pytest.raises(OrbmError):
        ctx.compute_bow(vid, [ids[0], ids[0]], 4, stride=10)    # a frame listed twice
    with pytest.raises(OrbmError):
        ctx.compute_bow(vid, ids, 4, stride=5)                  # stride below the frame size
    ctx.free_vocabulary(vid)
    with pytest.raises(OrbmError):
        ctx.compute_bow(vid, ids, 4, stride=10)                 # freed
    word, node = None, None
    vid = ctx.upload_vocabulary(good)
    word, node = ctx.compute_bow(vid, ids, 4, stride=10)        # the context still works
    assert (word >= 0).all() and (node == 0).all()
    ctx.free_vocabulary(vid)
    ctx.free_frames(ids)


@pytest.mark.gpu
def test_bow_score_parity(ctx, orc):
    """KeyFrameDatabase's similarity: DBoW2 L1 score between BowVectors, every pair of 48 vectors in one call, bit-equal
    doubles (the sum runs over the common words in ascending order on both sides) and equal common-word counts; empty
    vectors, a vector against itself, the 16384-word limit."""
    from orbslam2_amd.matcher import OrbmError
    from orbslam2_amd.views import BowVecData
    vecs = synth.bow_vectors(n_vec=46) + [BowVecData([], []), BowVecData([7], [1.0])]
    ids = ctx.upload_bowvecs(vecs)
    a, b = np.meshgrid(np.arange(len(vecs)), np.arange(len(vecs)), indexing="ij")
    a = a.ravel(); b = b.ravel()
    score, common = ctx.bow_score(ids[a], ids[b])
    want = [orc.bow_score(vecs[i], vecs[j]) for i, j in zip(a, b)]
    assert np.array_equal(score.view(np.uint64), np.array([w[0] for w in want], np.float64).view(np.uint64))
    assert np.array_equal(common, np.array([w[1] for w in want]))
    assert (common[a != b] > 100).sum() > 200 and np.all(np.abs(score[(a == b) & (a < 46)] - 1.0) < 1e-12)
    big = BowVecData(np.arange(16384, dtype=np.uint32) * 3, np.full(16384, 1.0 / 16384))
    bid = ctx.upload_bowvecs([big])
    s, c = ctx.bow_score([bid[0], ids[0]], [ids[0], bid[0]])
    assert (s[0], c[0]) == orc.bow_score(big, vecs[0]) and (s[1], c[1]) == orc.bow_score(vecs[0], big)
    with pytest.raises(OrbmError):
        ctx.upload_bowvecs([BowVecData(np.arange(16385, dtype=np.uint32), np.ones(16385))])
    with pytest.raises(OrbmError):
        ctx.upload_bowvecs([BowVecData([3, 3], [0.5, 0.5])])          # not strictly ascending
    ctx.free_bowvecs(bid)
    with pytest.raises(OrbmError):
        ctx.bow_score([bid[0]], [ids[0]])                             # freed
    ctx.free_bowvecs(ids)


def test_bow_search_follows_a_rebuilt_feature_vector(ctx, orc):
    """The matrix-core form of SearchByBoW reads a copy of each frame's descriptors in FeatureVector order, built the first
    time the frame is searched.  When ComputeBoW on the device replaces a frame's FeatureVector that copy is stale: the
    next search must follow the new grouping, and a frame freed and its handle reused must not inherit the old copy."""
    vocab_a = synth.make_vocabulary(seed=81, k=8, depth=3, flip=0.15)
    vocab_b = synth.make_vocabulary(seed=82, k=6, depth=4, flip=0.2)
    rng = np.random.default_rng(83)
    lm = synth.vocabulary_features(rng, vocab_a, 2500, flip=0.05)
    frames = []
    for k in range(4):
        sel = (300 * k + rng.integers(0, 1500, 900)) % 2500
        flags = np.where(rng.random(900) < 0.9, MP_PRESENT, 0).astype(np.uint8)
        frames.append(synth.make_frame(rng, lm, sel, with_fv=False, mp_flags=flags))
    ids = ctx.upload_frames(frames)
    pa, pb = np.meshgrid(np.arange(4), np.arange(4))
    pa, pb = pa.ravel(), pb.ravel()

    def regroup(f, vocab, levelsup):
        _, _, (f.fv_node, f.fv_off, f.fv_idx), _ = orc.transform(vocab, f.desc, levelsup, want_bow=True)

    def check(what):
        tot = 0
        for variant in (0, 1):
            counts, match = ctx.search_bow(variant, ids[pa], ids[pb], 0.75, True, stride=900)
            for t, (i, j) in enumerate(zip(pa, pb)):
                c, m, _ = orc.search_bow(variant, frames[i], frames[j], 0.75, True)
                assert counts[t] == c and np.array_equal(match[t, :len(m)], m), (what, variant, i, j)
                tot += c
        assert tot > 50, (what, tot)

    va = ctx.upload_vocabulary(vocab_a); vb = ctx.upload_vocabulary(vocab_b)
    live = list(ids)
    try:
        ctx.compute_bow(va, ids, 2, want_ids=False)
        for f in frames: regroup(f, vocab_a, 2)
        check("first grouping")                               # builds the FeatureVector-ordered copies
        ctx.compute_bow(vb, ids, 3, want_ids=False)           # other tree, other level: other nodes
        for f in frames: regroup(f, vocab_b, 3)
        check("after ComputeBoW with another vocabulary")
        ctx.compute_bow(va, ids[:2], 1, want_ids=False)       # only two of the frames regrouped
        for f in frames[:2]: regroup(f, vocab_a, 1)
        check("two of four frames regrouped")
        # handles are recycled: a new frame in an old slot gets its own copy
        ctx.free_frames(ids[2:]); live = list(ids[:2])
        other = [synth.make_frame(rng, lm, rng.integers(0, 2500, 700), mp_flags=np.full(700, MP_PRESENT, np.uint8)) for _ in range(2)]
        ids2 = ctx.upload_frames(other); live += list(ids2)
        assert set(ids2.tolist()) & set(ids[2:].tolist())
        counts, match = ctx.search_bow(1, np.array([ids2[0], ids2[1]]), np.array([ids2[1], ids2[0]]), 0.75, True, stride=700)
        for t, (i, j) in enumerate(((0, 1), (1, 0))):
            c, m, _ = orc.search_bow(1, other[i], other[j], 0.75, True)
            assert counts[t] == c and np.array_equal(match[t, :len(m)], m), ("recycled handle", i, j)
    finally:
        ctx.free_vocabulary(va); ctx.free_vocabulary(vb)
        ctx.free_frames(np.array(live, np.int32))


@pytest.mark.gpu
def test_relocalisation_chain_stays_on_the_device(ctx, orc):
    """Config 4 end to end without a host-built BoW structure: keyframes and a query frame uploaded with descriptors
    only -> ComputeBoW on the device (FeatureVector + BowVector) -> the query's BowVector scored against every keyframe
    -> the best-scoring keyframes matched with SearchByBoW(KF, F) in one batch.  Every stage equals the restated
    reference chain (transform -> L1 score -> SearchByBoW) on the host."""
    from orbslam2_amd.views import BowVecData
    vocab = synth.make_vocabulary(seed=71, k=10, depth=4, flip=0.15)
    rng = np.random.default_rng(72)
    lm = synth.vocabulary_features(rng, vocab, 6000, flip=0.05)
    n_kf, n = 40, 1200
    frames = []
    for k in range(n_kf + 1):                                  # keyframe k sees landmarks around 140 * k; the query sits at keyframe 17
        centre = 140 * (17 if k == n_kf else k)
        sel = (centre + rng.integers(0, 1600, n)) % 6000
        flags = np.where(rng.random(n) < 0.9, MP_PRESENT, 0).astype(np.uint8)
        frames.append(synth.make_frame(rng, lm, sel, with_fv=False, mp_flags=flags))
    ids = ctx.upload_frames(frames)
    vid = ctx.upload_vocabulary(vocab)
    try:
        (bv,) = ctx.compute_bow(vid, ids, 4, want_ids=False, want_bowvecs=True)
        score, common = ctx.bow_score(np.full(n_kf, bv[n_kf]), bv[:n_kf])
        host = []
        for f in frames:
            _, _, (f.fv_node, f.fv_off, f.fv_idx), (bw, bvv) = orc.transform(vocab, f.desc, 4, want_bow=True)
            host.append(BowVecData(bw, bvv))
        want = [orc.bow_score(host[n_kf], host[k]) for k in range(n_kf)]
        assert np.array_equal(score.view(np.uint64), np.array([w[0] for w in want]).view(np.uint64))
        assert np.array_equal(common, [w[1] for w in want])
        cand = np.argsort(-score, kind="stable")[:8]
        assert 17 in cand[:3]
        counts, match = ctx.search_bow(0, ids[cand], np.full(8, ids[n_kf]), 0.75, True, stride=n)
        for t, k in enumerate(cand):
            c, m, _ = orc.search_bow(0, frames[k], frames[n_kf], 0.75, True)
            assert counts[t] == c and np.array_equal(match[t, :n], m), k
        assert counts[list(cand).index(17)] > 50
        ctx.free_bowvecs(bv)
    finally:
        ctx.free_vocabulary(vid)
        ctx.free_frames(ids)


def test_one_call_searches_equal_the_batch_entry_points(ctx, orc):
    """orbm_search_proj_points_once / orbm_search_window_once (the query set travels with the call: one copy in, one out, one
    synchronisation) against the oracle, with and without slot flags, empty query sets and one-keypoint frames included."""
    rng = np.random.default_rng(77)
    for seed, n, nq, th, stereo in ((31, 2000, 1500, 3.0, False), (32, 700, 4000, 5.0, True), (33, 1, 10, 3.0, False), (34, 900, 0, 3.0, False),
                                    (35, 1500, 20000, 1.0, False)):
        f, q = synth.proj_instance(seed=seed, n=n, n_queries=nq, claimed_frac=0.1, obs_frac=0.7, stereo=stereo)
        fid = ctx.upload_frames([f])
        for use_slots in (False, True):
            sf = None
            if use_slots:
                sf = np.zeros(n, np.uint8); sf[rng.random(n) < 0.3] = MP_PRESENT | MP_OBS; sf[rng.random(n) < 0.1] = MP_PRESENT
            cnt, owner = ctx.search_proj_points_once(fid[0], q, th, 0.8, n, slot_flags=sf)
            wc, wo, _ = orc.search_proj_points(f, q, th, 0.8, slot_flags=sf)
            assert cnt == wc and np.array_equal(owner, wo), (seed, use_slots)
        ctx.free_frames(fid)
    for mode in (0, 1, 2):
        for wth, nq in ((7.0, 1500), (15.0, 1200), (7.0, 0)):
            f, q = synth.window_instance(990 + mode, mode, n=1500, n_queries=max(nq, 1), stereo=(mode == 0), th=wth)
            if nq == 0:
                from orbslam2_amd.views import WinQueries
                q = WinQueries(desc=q.desc[:0], u=q.u[:0], v=q.v[:0], radius=q.radius[:0], min_level=q.min_level[:0], max_level=q.max_level[:0],
                               ur=None if q.ur is None else q.ur[:0], angle=None if q.angle is None else q.angle[:0], flags=q.flags[:0])
            fid = ctx.upload_frames([f])
            dth = 100 if mode < 2 else 50
            cnt, owner = ctx.search_window_once(mode, fid[0], q, dth, mode < 2, 1500)
            wc, wo, _ = orc.search_window(mode, f, q, dth, mode < 2)
            assert cnt == wc and np.array_equal(owner, wo), (mode, wth, nq)
            ctx.free_frames(fid)
