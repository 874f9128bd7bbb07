"""GPU parity: a1 (DescriptorDistance), a15 (grid build) and BASELINE config 2 (brute-force 2-NN) through
the C ABI against the CPU oracle.  Bit-exact."""
import numpy as np
import pytest

from orbslam2_amd import synth

pytestmark = pytest.mark.gpu


def test_hamming_pairs_matches_swar(ctx, orc):
    rng = np.random.default_rng(1)
    a = synth.random_descriptors(rng, 5000)
    b = synth.random_descriptors(rng, 5000)
    b[:10] = a[:10]                      # distance 0
    b[10:20] = ~a[10:20]                 # distance 256
    got = ctx.hamming_pairs(a, b)
    want = np.array([orc.hamming256(a[i], b[i]) for i in range(len(a))], np.int32)
    assert np.array_equal(got, want)
    assert got[:10].max() == 0 and got[10:20].min() == 256


@pytest.mark.parametrize("n", [0, 1, 63, 1000, 2000, 8192, 9000])   # above 8192 the lists are ordered in global memory
def test_grid_build_matches_oracle(ctx, orc, n):
    rng = np.random.default_rng(100 + n)
    lm = synth.random_descriptors(rng, max(n, 1))
    # include keypoints outside the image bounds and exactly on cell borders
    x = rng.uniform(-20, 660, n).astype(np.float32)
    y = rng.uniform(-20, 500, n).astype(np.float32)
    if n >= 63:
        x[:20] = np.arange(20, dtype=np.float32) * 5.0   # multiples of half a cell: round() ties
        y[:20] = 15.0
    f = synth.make_frame(rng, lm, np.arange(n), x=x, y=y, with_fv=False)
    ids = ctx.upload_frames([f])
    off, idx = ctx.grid(ids[0], n)
    woff, widx = orc.build_grid(f)
    assert np.array_equal(off, woff)
    assert np.array_equal(idx, widx)
    ctx.free_frames(ids)


@pytest.mark.parametrize("variant", [0, 22, 24])       # 22 / 24: the large-batch forms (sign-free operands; last 64 bits deferred) on one pair
@pytest.mark.parametrize("na,nb", [(2000, 2000), (1, 1), (513, 77), (64, 1025), (700, 0), (0, 300)])
def test_bf2nn_parity(ctx, orc, na, nb, variant):
    from orbslam2_amd import _lib
    _lib.load().orbm__bf2nn_variant(variant)
    try:
        _bf2nn_parity(ctx, orc, na, nb)
    finally:
        _lib.load().orbm__bf2nn_variant(0)


def _bf2nn_parity(ctx, orc, na, nb):
    rng = np.random.default_rng(na * 7 + nb)
    n = max(na, nb, 1)
    a, b = synth.bf_pair(seed=na * 31 + nb, n=n)
    fa = synth.make_frame(rng, a.desc, np.arange(na), with_fv=False, p_flip=0.0)
    fb = synth.make_frame(rng, b.desc, np.arange(nb), with_fv=False, p_flip=0.0)
    ids = ctx.upload_frames([fa, fb])
    counts, match, bd, sd = ctx.bf2nn([ids[0]], [ids[1]], 0.75, 50, stride=max(na, 1))
    wc, wm, wbd, wsd, ncmp = orc.bf2nn(fa, fb, 0.75, 50)
    assert ncmp == na * nb
    assert counts[0] == wc
    assert np.array_equal(match[0, :na], wm)
    assert np.array_equal(bd[0, :na], wbd)
    assert np.array_equal(sd[0, :na], wsd)
    ctx.free_frames(ids)


def test_bf2nn_ties_first_wins_and_duplicates(ctx, orc):
    """Equal distances: the lower candidate index wins and the duplicate counts as second best, so the ratio
    test (ratio < 1) rejects the row (SURVEY 8c known answers)."""
    rng = np.random.default_rng(5)
    q = synth.random_descriptors(rng, 128)
    c = synth.random_descriptors(rng, 300)
    c[7] = q[3]; c[200] = q[3]            # two exact duplicates of query 3
    c[50] = q[9]; c[50, 0] ^= 1           # distance 1
    c[60] = q[9]; c[60, 5] ^= 16          # distance 1 again, later index
    c[100] = ~q[11]                       # distance 256 must never become "best"
    fa = synth.make_frame(rng, q, np.arange(128), with_fv=False, p_flip=0.0)
    fb = synth.make_frame(rng, c, np.arange(300), with_fv=False, p_flip=0.0)
    ids = ctx.upload_frames([fa, fb])
    counts, match, bd, sd = ctx.bf2nn([ids[0]], [ids[1]], 0.9, 50, stride=128)
    wc, wm, wbd, wsd, _ = orc.bf2nn(fa, fb, 0.9, 50)
    assert np.array_equal(match[0], wm) and np.array_equal(bd[0], wbd) and np.array_equal(sd[0], wsd)
    assert bd[0, 3] == 0 and sd[0, 3] == 0 and match[0, 3] == -1
    assert bd[0, 9] == 1 and sd[0, 9] == 1 and match[0, 9] == -1
    # with ratio > 1 the tie is accepted and the FIRST index must be reported
    counts, match, _, _ = ctx.bf2nn([ids[0]], [ids[1]], 1.5, 50, stride=128)
    _, wm, _, _, _ = orc.bf2nn(fa, fb, 1.5, 50)
    assert np.array_equal(match[0], wm)
    assert match[0, 3] == -1 and match[0, 9] == 50   # 0 < 1.5*0 is false; 1 < 1.5*1 holds
    ctx.free_frames(ids)


def test_bf2nn_batch_many_pairs(ctx, orc):
    kfs = synth.keyframe_pool(seed=77, n_kf=6, n=500, window=900)
    ids = ctx.upload_frames(kfs)
    pa, pb = np.triu_indices(6, 1)
    counts, match, bd, sd = ctx.bf2nn(ids[pa], ids[pb], 0.75, 50, stride=500)
    for k, (i, j) in enumerate(zip(pa, pb)):
        wc, wm, wbd, wsd, _ = orc.bf2nn(kfs[i], kfs[j], 0.75, 50)
        assert counts[k] == wc, (i, j)
        assert np.array_equal(match[k], wm) and np.array_equal(bd[k], wbd) and np.array_equal(sd[k], wsd)
    ctx.free_frames(ids)


@pytest.mark.parametrize("variant", [1, 2, 3, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 22, 24])
def test_bf2nn_kernel_variants_give_the_same_matches(ctx, orc, variant):
    """The vector-ALU kernel (11) and its tile-shape variants, the track-everything variant, the opt-in early-abandon
    variant (8: the last 64 bits of a comparison are skipped once the first 192 already reach the tracking bound), the
    int8 MFMA form (9), the FP4 MFMA form fed from pre-expanded images through LDS (10) and the FP4 MFMA form fed from the
    raw rows with two / four / one query tiles per wave (12 / 13 / 14; the default picks among these by batch size, which
    every other test of this file runs), the four-tile form with +-1 operands (17) and with sign-free operands (22: candidate
    bits complemented as 0 / 1, query bits as 0 / 2, row popcounts as the accumulators' initial values; what large batches
    run) all return the reference's matches, with and without the raw distances, across
    thresholds and ratios."""
    from orbslam2_amd import _lib
    L = _lib.load()
    kfs = synth.keyframe_pool(seed=78, n_kf=5, n=700, window=1000) + contended_frames(79)
    ids = ctx.upload_frames(kfs)
    pa, pb = np.meshgrid(np.arange(len(kfs)), np.arange(len(kfs)))
    pa, pb = pa.ravel(), pb.ravel()
    try:
        L.orbm__bf2nn_variant(variant)
        for ratio, th in ((0.75, 50), (0.6, 50), (0.95, 100), (1.5, 30), (0.75, 0), (0.9, 256)):
            for want_dist in (False, True):
                counts, match, bd, sd = ctx.bf2nn(ids[pa], ids[pb], ratio, th, stride=700, want_dist=want_dist)
                for k, (i, j) in enumerate(zip(pa, pb)):
                    wc, wm, wbd, wsd, _ = orc.bf2nn(kfs[i], kfs[j], ratio, th)
                    assert counts[k] == wc and np.array_equal(match[k, :kfs[i].n], wm), (variant, ratio, th, want_dist, i, j)
                    if want_dist:
                        assert np.array_equal(bd[k, :kfs[i].n], wbd) and np.array_equal(sd[k, :kfs[i].n], wsd)
    finally:
        L.orbm__bf2nn_variant(0)
    ctx.free_frames(ids)


@pytest.mark.parametrize("variant", [0, 12, 13, 14, 22, 24])
def test_bf2nn_tile_boundaries_mixed_sizes(ctx, orc, variant):
    """Frames of every size around the 32-row tile and the 128 / 256 / 512-row workgroup boundaries, mixed in ONE launch
    (a wave's number of live query tiles, the last candidate block and the rows between a frame's n and the stride all
    differ from pair to pair), through the raw-row MFMA forms with one / two / four query tiles per wave and the default
    choice, with and without raw distances; padding entries of the match rows must read -1."""
    from orbslam2_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(4242)
    lm = synth.random_descriptors(rng, 700)
    sizes = [1, 2, 31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 128, 129, 255, 257, 383, 385, 511, 512, 513, 640]
    frames = [synth.make_frame(rng, lm, rng.permutation(700)[:n], with_fv=False, p_flip=0.06) for n in sizes]
    ids = ctx.upload_frames(frames)
    k = len(sizes)
    pa, pb = np.meshgrid(np.arange(k), np.arange(k))
    pa, pb = pa.ravel(), pb.ravel()
    stride = 656        # not a multiple of 32, above the largest frame
    try:
        L.orbm__bf2nn_variant(variant)
        for want_dist in (False, True):
            counts, match, bd, sd = ctx.bf2nn(ids[pa], ids[pb], 0.8, 60, stride=stride, want_dist=want_dist)
            for q, (i, j) in enumerate(zip(pa, pb)):
                wc, wm, wbd, wsd, _ = orc.bf2nn(frames[i], frames[j], 0.8, 60)
                n = frames[i].n
                assert counts[q] == wc and np.array_equal(match[q, :n], wm), (variant, want_dist, sizes[i], sizes[j])
                assert (match[q, n:] == -1).all(), (variant, want_dist, sizes[i], sizes[j])
                if want_dist:
                    assert np.array_equal(bd[q, :n], wbd) and np.array_equal(sd[q, :n], wsd), (variant, sizes[i], sizes[j])
    finally:
        L.orbm__bf2nn_variant(0)
    ctx.free_frames(ids)


def contended_frames(seed):
    """Frames re-observing a small landmark pool with little noise: many near-ties around the thresholds."""
    rng = np.random.default_rng(seed)
    lm = synth.random_descriptors(rng, 40)
    out = []
    for p_flip, n in ((0.02, 300), (0.1, 450), (0.2, 200)):
        out.append(synth.make_frame(rng, lm, rng.integers(0, 40, n), with_fv=False, p_flip=p_flip))
    return out


def test_errors_are_loud(ctx):
    from orbslam2_amd.matcher import OrbmError
    with pytest.raises(OrbmError):
        ctx.bf2nn([12345], [0], 0.75, 50, stride=10)


@pytest.mark.parametrize("ratio,th", [(0.75, 50), (0.6, 50), (0.9, 100), (1.5, 50), (0.75, 300), (0.0, 50), (0.75, -1)])
def test_bf2nn_matches_only_path(ctx, orc, ratio, th):
    """Without the raw-distance outputs the kernel caps its tracking at the acceptance bound; matches and
    counts must not change (2000x2000 is BASELINE config 2; 515x1031 exercises the tile tail)."""
    for (na, nb, seed) in ((2000, 2000, 1), (515, 1031, 2)):
        rng = np.random.default_rng(seed)
        a, b = synth.bf_pair(seed=seed, n=max(na, nb))
        fa = synth.make_frame(rng, a.desc, np.arange(na), with_fv=False, p_flip=0.0)
        fb = synth.make_frame(rng, b.desc, np.arange(nb), with_fv=False, p_flip=0.0)
        ids = ctx.upload_frames([fa, fb])
        counts, match, _, _ = ctx.bf2nn([ids[0]], [ids[1]], ratio, th, stride=na, want_dist=False)
        wc, wm, _, _, _ = orc.bf2nn(fa, fb, ratio, th)
        assert counts[0] == wc
        assert np.array_equal(match[0], wm)
        ctx.free_frames(ids)


def test_bf2nn_bench_size_properties(ctx, orc):
    """The bench's own workload (4096 pairs of 2000 x 2000 from a 96-keyframe store, one launch) through properties that
    need no oracle at that size, plus the oracle on a sample of the pairs:
    * batch invariance: the same pairs in four launches, and in reversed order, give the same rows;
    * a keyframe against itself: every descriptor finds itself at distance 0;
    * every accepted match satisfies the reference's accept rule on the returned best / second distances, every
      rejected row violates it; second >= best; match count = number of accepted rows;
    * the two smallest distances of 200 random (pair, row) samples recomputed on the host with a lookup-table popcount,
      and the reported match = the first index that reaches the minimum;
    * 12 sampled pairs bit-equal to the oracle."""
    kfs = synth.keyframe_pool(seed=2024, n_kf=96, n=2000, window=3000)
    ids = ctx.upload_frames(kfs)
    rng = np.random.default_rng(9)
    P = 4096
    pa = rng.integers(0, 96, P); pb = rng.integers(0, 96, P)
    pa[:96] = np.arange(96); pb[:96] = np.arange(96)                    # self pairs
    counts, match, bd, sd = ctx.bf2nn(ids[pa], ids[pb], 0.75, 50, stride=2000)
    # batch invariance
    for lo in range(0, P, 1024):
        c2, m2, b2, s2 = ctx.bf2nn(ids[pa[lo:lo + 1024]], ids[pb[lo:lo + 1024]], 0.75, 50, stride=2000)
        assert np.array_equal(c2, counts[lo:lo + 1024]) and np.array_equal(m2, match[lo:lo + 1024])
        assert np.array_equal(b2, bd[lo:lo + 1024]) and np.array_equal(s2, sd[lo:lo + 1024])
    c3, m3, _, _ = ctx.bf2nn(ids[pa[::-1]], ids[pb[::-1]], 0.75, 50, stride=2000)
    assert np.array_equal(c3[::-1], counts) and np.array_equal(m3[::-1], match)
    # self pairs
    assert (bd[:96] == 0).all()
    self_rows = match[:96]
    ok = self_rows == np.arange(2000)[None, :]
    assert ok[self_rows >= 0].all() and (self_rows >= 0).mean() > 0.99  # a rejected row = an exact duplicate descriptor in the frame
    # accept rule on the returned distances: best <= th and (float)best < ratio * (float)second
    accept = (bd <= 50) & (bd.astype(np.float32) < np.float32(0.75) * sd.astype(np.float32))
    assert np.array_equal(accept, match >= 0)
    assert (sd >= bd).all() and np.array_equal(counts, (match >= 0).sum(axis=1))
    assert ((match >= -1) & (match < 2000)).all()
    # row minima recomputed on the host for a sample
    lut = np.array([bin(i).count("1") for i in range(256)], np.int32)
    for _ in range(200):
        p = int(rng.integers(0, P)); r = int(rng.integers(0, 2000))
        d = lut[kfs[pa[p]].desc[r][None, :] ^ kfs[pb[p]].desc].sum(axis=1)
        order = np.argsort(d, kind="stable")
        assert d[order[0]] == bd[p, r] and d[order[1]] == sd[p, r]
        if match[p, r] >= 0:
            assert match[p, r] == order[0]
    # the oracle on a sample of pairs
    for p in rng.integers(96, P, 12):
        wc, wm, wbd, wsd, _ = orc.bf2nn(kfs[pa[p]], kfs[pb[p]], 0.75, 50)
        assert counts[p] == wc and np.array_equal(match[p], wm) and np.array_equal(bd[p], wbd) and np.array_equal(sd[p], wsd)
    ctx.free_frames(ids)


@pytest.mark.parametrize("variant", [0, 17, 22, 24])
def test_bf2nn_every_distance_value_is_exact(ctx, orc, variant):
    """Every Hamming distance 0..256 through the matrix-core forms (distance = (256 - dot) / 2 of +-1 images; 22: the
    sign-free operands, result = distance + popcount(query) on top of the candidate row's popcount): for each d a
    pair whose candidates sit at exactly d, min(d + 3, 256) and 256 bits from the query (bit positions chosen at random,
    so every k step of the product takes part); best / second come back as exactly those values, and the accept decision
    flips where the reference's does (th = d, th = d - 1; the ratio boundary)."""
    rng = np.random.default_rng(123)
    frames, want = [], []
    for d in range(257):
        q = synth.random_descriptors(rng, 1)
        def at(dist):
            bits = np.unpackbits(q[0]).copy()
            flip = rng.permutation(256)[:dist]
            bits[flip] ^= 1
            return np.packbits(bits)
        d2 = min(d + 3, 256)
        cands = np.stack([at(256), at(d2), at(d), at(d2)])       # best at index 2; the second-best value appears twice
        frames.append(synth.make_frame(rng, q, np.arange(1), with_fv=False, p_flip=0.0))
        frames.append(synth.make_frame(rng, cands, np.arange(4), with_fv=False, p_flip=0.0))
        want.append((d, d2))
    ids = ctx.upload_frames(frames)
    fa, fb = ids[0::2], ids[1::2]
    from orbslam2_amd import _lib
    L = _lib.load()
    try:
        L.orbm__bf2nn_variant(variant)
        for ratio, th in ((0.75, 50), (1.0, 256), (0.99, 100), (2.0, 0)):
            counts, match, bd, sd = ctx.bf2nn(fa, fb, ratio, th, stride=1)
            c2, m2, _, _ = ctx.bf2nn(fa, fb, ratio, th, stride=1, want_dist=False)
            assert np.array_equal(counts, c2) and np.array_equal(match, m2)
            for d, (wb, ws) in enumerate(want):
                assert (bd[d, 0], sd[d, 0]) == (wb, ws), (d, bd[d, 0], sd[d, 0])
                accept = wb <= th and np.float32(wb) < np.float32(ratio) * np.float32(ws)
                assert match[d, 0] == (2 if accept else -1) and counts[d] == int(accept), (ratio, th, d)
                wc, wm, wbd, wsd, _ = orc.bf2nn(frames[2 * d], frames[2 * d + 1], ratio, th)
                assert wc == counts[d] and wm[0] == match[d, 0] and wbd[0] == wb and wsd[0] == ws
    finally:
        L.orbm__bf2nn_variant(0)
    ctx.free_frames(ids)


@pytest.mark.parametrize("variant", [0, 17, 22, 24])
def test_bf2nn_extreme_popcounts(ctx, orc, variant):
    """Descriptors of every weight: all zeros, all ones, single bits, one zero bit, sparse and dense rows on both sides (in
    the sign-free form a result is distance + popcount(query) on top of popcount(candidate): 0 .. 768), 70 x 70 rows so
    that candidate blocks mix the extremes, with raw distances and without, thresholds up to 256."""
    rng = np.random.default_rng(31)
    rows = [np.zeros(32, np.uint8), np.full(32, 255, np.uint8)]
    for k in (0, 7, 100, 255):
        one = np.zeros(256, np.uint8); one[k] = 1
        rows.append(np.packbits(one)); rows.append(np.packbits(1 - one))
    for p in (0.02, 0.1, 0.3, 0.5, 0.7, 0.9, 0.98):
        for _ in range(8):
            rows.append(np.packbits((rng.random(256) < p).astype(np.uint8)))
    rows = np.stack(rows)
    from orbslam2_amd import _lib
    L = _lib.load()
    fa = synth.make_frame(rng, rows, rng.permutation(len(rows)), with_fv=False, p_flip=0.0)
    fb = synth.make_frame(rng, rows, rng.permutation(len(rows)), with_fv=False, p_flip=0.0)
    fc = synth.make_frame(rng, rows, rng.permutation(len(rows)), with_fv=False, p_flip=0.01)
    ids = ctx.upload_frames([fa, fb, fc])
    fr = [fa, fb, fc]
    try:
        L.orbm__bf2nn_variant(variant)
        for ratio, th in ((0.75, 50), (1.0, 256), (1.5, 200), (0.9, 128)):
            for want_dist in (True, False):
                pa, pb = [0, 1, 2, 0, 2], [1, 0, 0, 2, 2]
                counts, match, bd, sd = ctx.bf2nn(ids[pa], ids[pb], ratio, th, stride=len(rows), want_dist=want_dist)
                for k, (i, j) in enumerate(zip(pa, pb)):
                    wc, wm, wbd, wsd, _ = orc.bf2nn(fr[i], fr[j], ratio, th)
                    assert counts[k] == wc and np.array_equal(match[k], wm), (variant, ratio, th, want_dist, i, j)
                    if want_dist:
                        assert np.array_equal(bd[k], wbd) and np.array_equal(sd[k], wsd), (variant, ratio, th, i, j)
    finally:
        L.orbm__bf2nn_variant(0)
    ctx.free_frames(ids)


def test_stray_diagnostic_environment_cannot_change_results():
    """The release library has no form that returns wrong matches: with the timing-diagnostic switches of the `make diag`
    build set in the environment (ORBM_MFMA_DIAG) and a diagnostic variant requested through the knob, the LDS-staged MFMA
    form still returns the oracle's matches.  Runs in a fresh process: the library reads its environment once."""
    import subprocess
    import sys
    code = r'''
import numpy as np
from oracle import oracle_ffi as orc
from orbslam2_amd import synth, _lib
from orbslam2_amd.matcher import Context
ctx = Context(0)
L = _lib.load()
kfs = synth.keyframe_pool(seed=77, n_kf=3, n=700, window=1100)
ids = ctx.upload_frames(kfs)
for variant in (0, 4, 18, 19, 20):
    L.orbm__bf2nn_variant(variant)
    counts, match, bd, sd = ctx.bf2nn([ids[0], ids[1]], [ids[1], ids[2]], 0.75, 50, stride=700, want_dist=False)
    for k, (i, j) in enumerate(((0, 1), (1, 2))):
        wc, wm, _, _, _ = orc.bf2nn(kfs[i], kfs[j], 0.75, 50)
        assert counts[k] == wc and np.array_equal(match[k], wm), (variant, k)
L.orbm__bf2nn_variant(0)
assert not hasattr(L, "orbm__window_cost_exact")
ctx.close()
print("ok")
'''
    import os
    env = dict(os.environ, ORBM_MFMA_DIAG="1", ORBMATCH_BF2NN_KERNEL="lds")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]


def test_kernel_timing_records_the_matching_kernel(ctx, orc):
    """orbm_kernel_timing / orbm_kernel_times (bench.py's roofline figure): one event pair per brute-force launch, read back after
    a synchronisation; results are untouched and the recording stops at the capacity asked for."""
    import ctypes as C
    kfs = synth.keyframe_pool(seed=77, n_kf=2, n=500)
    ids = ctx.upload_frames(kfs)
    L = ctx._L
    assert L.orbm_kernel_timing(ctx._h, 2) == 0
    for _ in range(3):                                       # the third launch finds no event pair left: not recorded, not an error
        c, m, bd, sd = ctx.bf2nn([ids[0]], [ids[1]], 0.75, 50, stride=500)
    wc, wm, wb, ws, _ = orc.bf2nn(kfs[0], kfs[1], 0.75, 50)
    assert c[0] == wc and np.array_equal(m[0], wm)
    buf = (C.c_float * 4)()
    assert L.orbm_kernel_times(ctx._h, buf, 4) == 2
    assert 0.0 < buf[0] < 50.0 and 0.0 < buf[1] < 50.0
    assert L.orbm_kernel_times(ctx._h, buf, 4) == 0          # reading starts the recording over
    assert L.orbm_kernel_timing(ctx._h, 0) == 0
    ctx.free_frames(ids)
