"""Error behaviour of the C ABI (through the ctypes mirror): bad arguments are refused with a message, nothing is
launched, and the context stays usable."""
import numpy as np
import pytest

from orbslam2_amd import synth
from orbslam2_amd.views import FrameData

pytestmark = pytest.mark.gpu


def test_bad_arguments_are_refused_and_context_survives(ctx, orc):
    from orbslam2_amd.matcher import OrbmError
    kfs = synth.keyframe_pool(seed=5, n_kf=2, n=200)
    ids = ctx.upload_frames(kfs)
    f, q = synth.proj_instance(seed=6, n=200, n_queries=300)
    fw, qw = synth.window_instance(7, 1, n=200, n_queries=100)
    pid = ctx.upload_proj_queries([q]); wid = ctx.upload_win_queries([qw])
    bad_calls = [
        lambda: ctx.bf2nn([ids[0]], [9999], 0.75, 50, stride=200),                       # unknown frame id
        lambda: ctx.bf2nn([ids[0]], [-1], 0.75, 50, stride=200),
        lambda: ctx.bf2nn([ids[0]], [ids[1]], 0.75, 50, stride=10),                      # stride below the row count
        lambda: ctx.search_bow(2, [ids[0]], [ids[1]], 0.75, True, stride=200),           # variant out of range
        lambda: ctx.search_bow(0, [ids[0]], [ids[1]], 0.75, True, stride=5),
        lambda: ctx.search_proj_points([ids[0]], wid, 3.0, 0.8, stride=200),             # window query set given to the a6 entry
        lambda: ctx.search_window(1, [ids[0]], pid, 64, True, stride=200),               # and the other way round
        lambda: ctx.search_window(7, [ids[0]], wid, 64, True, stride=200),               # mode out of range
        lambda: ctx.search_window_free([ids[0]], wid, 50, True, stride_q=100),           # chi-square gate without ur
        lambda: ctx.search_window_free([ids[0]], wid, 50, False, stride_q=10),           # stride below the query count
        lambda: ctx.stereo_match([ids[0]], [ids[1]], [40.0], stride=10),
    ]
    for k, call in enumerate(bad_calls):
        with pytest.raises(OrbmError) as e:
            call()
        assert str(e.value), k
    # malformed frames are refused at upload
    rng = np.random.default_rng(0)
    lm = synth.random_descriptors(rng, 10)
    g = synth.make_frame(rng, lm, np.arange(10))
    g.octave = g.octave.copy(); g.octave[3] = 9                                          # outside [0, n_levels)
    with pytest.raises(OrbmError):
        ctx.upload_frames([g])
    g = synth.make_frame(rng, lm, np.arange(10))
    g.fv_idx = g.fv_idx.copy(); g.fv_idx[0] = 10                                         # FeatureVector index out of range
    with pytest.raises(OrbmError):
        ctx.upload_frames([g])
    # a freed id is dead
    ctx.free_frames([ids[1]])
    with pytest.raises(OrbmError):
        ctx.bf2nn([ids[0]], [ids[1]], 0.75, 50, stride=200)
    # ... and the context still works
    c, m, bd, sd = ctx.bf2nn([ids[0]], [ids[0]], 0.75, 50, stride=200)
    wc, wm, wb, ws, _ = orc.bf2nn(kfs[0], kfs[0], 0.75, 50)
    assert c[0] == wc and np.array_equal(m[0], wm)
    ctx.free_frames([ids[0]]); ctx.free_queries(pid); ctx.free_queries(wid)


@pytest.mark.gpu
def test_no_device_memory_growth_over_upload_match_free_cycles(ctx, orc):
    """300 cycles of upload -> brute-force match (builds the +-1 images) -> ComputeBoW with BowVectors -> score -> free
    must not grow the device footprint: arenas, images, BowVector arenas and table slots are all recycled.  (A leak of
    one cycle's 6 MB would show as 1.8 GB; released arenas may stay pooled, bounded at 256 MB.)"""
    import torch
    from orbslam2_amd import synth
    vocab = synth.make_vocabulary(seed=3, k=6, depth=3)
    vid = ctx.upload_vocabulary(vocab)
    rng = np.random.default_rng(4)
    lm = synth.random_descriptors(rng, 3000)
    frames = [synth.make_frame(rng, lm, rng.integers(0, 3000, 1500), with_fv=False) for _ in range(6)]

    def cycle():
        ids = ctx.upload_frames(frames)
        counts, match, _, _ = ctx.bf2nn(ids[:3], ids[3:], 0.75, 50, stride=1500, want_dist=False)
        (bv,) = ctx.compute_bow(vid, ids, 4, want_ids=False, want_bowvecs=True)
        score, _ = ctx.bow_score(bv[:3], bv[3:])
        ctx.free_bowvecs(bv)
        ctx.free_frames(ids)
        return int(counts.sum()), float(score.sum())

    first = cycle()
    for _ in range(20):
        cycle()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(300):
        assert cycle() == first
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    ctx.free_vocabulary(vid)
    assert free0 - free1 < 300 * 1024 * 1024, (free0, free1)


def test_round3_entry_points_refuse_bad_arguments(ctx, orc):
    """orbm_search_*_once, orbm_allpairs: bad arguments are errors with a message, empty inputs are not, and the context
    works afterwards."""
    from orbslam2_amd.matcher import Context, OrbmError
    f, q = synth.proj_instance(seed=16, n=300, n_queries=200)
    fw, qw = synth.window_instance(17, 0, n=300, n_queries=150, th=7.0, stereo=True)
    ids = ctx.upload_frames([f, fw])
    bad_calls = [
        lambda: ctx.search_proj_points_once(9999, q, 3.0, 0.8, 300),                      # unknown frame
        lambda: ctx.search_window_once(5, ids[1], qw, 100, True, 300),                    # mode out of range
        lambda: ctx.search_window_once(0, -3, qw, 100, True, 300),
        lambda: Context.allpairs_multi([ctx], [ids], 7, [0], [1], None, 0.75, 50, True, 300, 20, 64),      # kind out of range
    ]
    for k, call in enumerate(bad_calls):
        with pytest.raises(OrbmError) as e:
            call()
        assert str(e.value), k
    # an empty query set is a search with no matches, not an error
    import copy
    q0 = copy.copy(q)
    for name in ("desc", "proj_x", "proj_y", "proj_xr", "level", "view_cos"):
        setattr(q0, name, getattr(q, name)[:0])
    if hasattr(q0, "flags") and q0.flags is not None:
        q0.flags = q0.flags[:0]
    cnt, owner = ctx.search_proj_points_once(ids[0], q0, 3.0, 0.8, 300)
    assert cnt == 0 and (owner == -1).all()
    # ... and the context still gives the oracle's answer
    cnt, owner = ctx.search_proj_points_once(ids[0], q, 3.0, 0.8, 300)
    wc, wo, _ = orc.search_proj_points(f, q, 3.0, 0.8)
    assert cnt == wc and np.array_equal(owner[:f.n], wo)
    ctx.free_frames(ids)
