"""Compact match records of config 5 on the GPU: orbm_pack_records against the numpy statement of the
format, orbm_allpairs_batch (match + pack in one call) against the oracle's matches for the three kinds of search."""
import os
import sys

import numpy as np
import pytest

from orbslam2_amd import allpairs, synth
from orbslam2_amd.matcher import OrbmError

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import packref  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_pairs,stride,threshold", [(0, 8, 1), (1, 1, 1), (3, 70, 0), (1500, 130, 20), (5000, 64, 15), (37, 2000, 20)])
def test_pack_records_matches_the_format_definition(ctx, n_pairs, stride, threshold):
    rng = np.random.default_rng(n_pairs + stride)
    match = np.full((n_pairs, stride), -1, np.int32)
    counts = np.zeros(n_pairs, np.int32)
    for k in range(n_pairs):
        c = int(rng.integers(0, min(stride, 60) + 1)) if rng.random() < 0.5 else 0
        rows = rng.choice(stride, c, replace=False)
        match[k, rows] = rng.integers(0, 65536, c)
        counts[k] = c
    pid = rng.integers(0, 1 << 32, n_pairs, dtype=np.uint64).astype(np.uint32)
    if n_pairs > 10:
        pid[rng.choice(n_pairs, 5, replace=False)] = packref.PAD_ID          # padding entries: n = 0, never listed
    want_s, want_l, want_t = packref.pack(pid, counts, match, threshold)
    summ, lists, totals = ctx.pack_records(pid, counts, match.reshape(n_pairs, stride), threshold, list_cap=max(len(want_l), 1) + 7)
    assert tuple(totals) == want_t
    assert np.array_equal(summ.view(np.uint64), want_s.view(np.uint64)) and np.array_equal(lists, want_l)
    if len(want_l) > 4:                                                       # a short buffer cuts the lists, the totals say so
        summ, lists, totals = ctx.pack_records(pid, counts, match, threshold, list_cap=len(want_l) // 2)
        assert tuple(totals) == want_t and np.array_equal(lists, want_l[: len(want_l) // 2])
        assert np.array_equal(summ.view(np.uint64), want_s.view(np.uint64))
    if n_pairs:                                                               # pair_id NULL = position in the batch
        summ, _, _ = ctx.pack_records(None, counts, match, threshold, list_cap=len(want_l) + 1)
        assert np.array_equal(summ["pair_id"], np.arange(n_pairs))


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_allpairs_batch_against_the_oracle(ctx, orc, kind):
    n, n_kf = 700, 14
    kfs = synth.keyframe_pool(seed=synth.SEED0 + 50 + kind, n_kf=n_kf, n=n, window=1200, mp_valid_frac=0.9)
    kfs[3] = synth.keyframe_pool(seed=9, n_kf=1, n=450, window=1200)[0]      # a shorter frame: rows beyond its n read "no match"
    ids = ctx.upload_frames(kfs)
    a, b = allpairs.shard_pairs(n_kf, 0, 1, tile=4)
    pid = allpairs.pair_ids(a, b, n_kf)
    threshold = 20
    summ, lists, totals = ctx.allpairs_batch(kind, ids[a], ids[b], pid, 0.75, 50, True, n, threshold, list_cap=len(a) * n)
    off, n_acc = 0, 0
    for k in range(len(a)):
        if kind == 0:
            wc, wm = orc.bf2nn(kfs[a[k]], kfs[b[k]], 0.75, 50)[:2]
        else:
            wc, wm, _ = orc.search_bow(1 if kind == 1 else 0, kfs[a[k]], kfs[b[k]], 0.75, True)
        assert summ["pair_id"][k] == pid[k] and summ["n"][k] == wc and summ["pad"][k] == 0, (kind, k)
        if wc >= threshold:
            rows = np.nonzero(wm >= 0)[0]
            e = lists[off:off + wc]
            assert np.array_equal(e & 0xFFFF, rows) and np.array_equal(e >> 16, wm[rows]), (kind, k)
            off += wc; n_acc += 1
    assert tuple(totals) == (n_acc, off) and len(lists) == off and n_acc >= 5
    with pytest.raises(OrbmError):
        ctx.allpairs_batch(kind, ids[a], ids[b], pid, 0.75, 50, True, 100, threshold, list_cap=10)   # stride below a frame's n
    ctx.free_frames(ids)


@pytest.mark.parametrize("kind,n_ctx", [(0, 2), (1, 3), (2, 2)])
def test_allpairs_over_several_contexts_equals_one_context(ctx, kind, n_ctx):
    """orbm_allpairs: the pair list cut over several contexts -- here all on the test box's one GPU, each with
    its own replica of the store, handles differing between contexts -- returns exactly the records one context returns."""
    from orbslam2_amd.matcher import Context
    n, n_kf = 600, 11
    kfs = synth.keyframe_pool(seed=synth.SEED0 + 70 + kind, n_kf=n_kf, n=n, window=1000, mp_valid_frac=0.9)
    ids = ctx.upload_frames(kfs)
    a, b = allpairs.shard_pairs(n_kf, 0, 1, tile=4)
    pid = allpairs.pair_ids(a, b, n_kf)
    want = ctx.allpairs_batch(kind, ids[a], ids[b], pid, 0.75, 50, True, n, 20, list_cap=len(a) * n)
    others = [Context(0) for _ in range(n_ctx - 1)]
    try:
        handles = [ids]
        for k, c in enumerate(others):
            c.upload_frames(kfs[: k + 1])                       # shift the handle numbering of this context
            handles.append(c.upload_frames(kfs))
        got = Context.allpairs_multi([ctx] + others, handles, kind, a, b, pid, 0.75, 50, True, n, 20, list_cap=len(a) * n)
        assert np.array_equal(got[0].view(np.uint64), want[0].view(np.uint64))
        assert np.array_equal(got[1], want[1]) and tuple(got[2]) == tuple(want[2]) and want[2][0] > 0
        # a short list buffer: cut at list_cap, totals still say how many there were
        cap = int(want[2][1]) // 2
        cut = Context.allpairs_multi([ctx] + others, handles, kind, a, b, pid, 0.75, 50, True, n, 20, list_cap=cap)
        assert tuple(cut[2]) == tuple(want[2]) and np.array_equal(cut[1], want[1][:cap])
        with pytest.raises(OrbmError):
            Context.allpairs_multi([ctx] + others, handles, 7, a, b, pid, 0.75, 50, True, n, 20, list_cap=16)
    finally:
        for c in others:
            c.close()
    ctx.free_frames(ids)
