"""BASELINE.json configs 3, 4 and 5 at their STATED sizes, HIP path vs the CPU oracle, bit for bit.

  C3  SearchByProjection(F, MapPoints, th): 50 000 projected map points -> one 2000-keypoint frame
      (ORBmatcher.cc:57-166), th in {1, 3, 5}; and bench.py's own launch shape (256 instances, 64 distinct query
      sets), which is the input that reaches the 2048-query super-chunks and the big-batch row layout.
  C4  SearchByBoW: ONE frame against 64 candidate keyframes x 2000 features (ORBmatcher.cc:218-385, the loop of
      Tracking.cc:1778-1801 / LoopClosing.cc:433-477), both variants, all 64 pairs checked.
  C5  all pairs of a keyframe store (ORBmatcher.cc:703-854 per pair): the full upper triangle over 320 keyframes x
      2000 features through the tiling of orbslam2_amd/allpairs.py and the batching of scripts/c5_allpairs.py,
      200 random pairs compared with the oracle, per-pair counts invariant under the batch size.
"""
import numpy as np
import pytest

from orbslam2_amd import allpairs, synth
from orbslam2_amd.views import ProjQueries

pytestmark = pytest.mark.gpu

N = 2000


@pytest.mark.parametrize("th", [1.0, 3.0, 5.0])
def test_c3_50k_map_points_into_one_frame(ctx, orc, th):
    f, q = synth.proj_instance(seed=synth.SEED0 + 3, n=N, n_queries=50000)
    fid = ctx.upload_frames([f]); qid = ctx.upload_proj_queries([q])
    counts, owner = ctx.search_proj_points(fid, qid, th, 0.8, stride=N)
    wc, wo, _ = orc.search_proj_points(f, q, th, 0.8)
    assert counts[0] == wc and np.array_equal(owner[0], wo)
    assert wc > 1000
    ctx.free_frames(fid); ctx.free_queries(qid)


def test_c3_bench_launch_shape_256_instances(ctx, orc):
    """bench.py's config-3 launch: 256 instances, 64 distinct permutations of the 50k query set, 8 frame copies."""
    import torch
    dev = torch.device("cuda", 0)
    n_inst, n_sets, NQ = 256, 64, 50000
    f, q = synth.proj_instance(seed=synth.SEED0 + 3, n=N, n_queries=NQ)
    prng = np.random.default_rng(0)
    sets = []
    for _ in range(n_sets):
        p = prng.permutation(q.n)
        sets.append(ProjQueries(desc=q.desc[p], proj_x=q.proj_x[p], proj_y=q.proj_y[p], proj_xr=q.proj_xr[p], level=q.level[p],
                                view_cos=q.view_cos[p], flags=q.flags[p]))
    fid = ctx.upload_frames([f] * 8); qid = ctx.upload_proj_queries(sets)
    tfr = torch.tensor(np.resize(fid, n_inst), device=dev); tqs = torch.tensor(np.resize(qid, n_inst), device=dev)
    owner = torch.full((n_inst, N), -7, dtype=torch.int32, device=dev); cnt = torch.full((n_inst,), -7, dtype=torch.int32, device=dev)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        ctx.search_proj_points_dev(n_inst, tfr.data_ptr(), tqs.data_ptr(), 3.0, 0.8, None, N, owner.data_ptr(), cnt.data_ptr())
        torch.cuda.synchronize()
    finally:
        ctx.use_own_stream()
    owner = owner.cpu().numpy(); cnt = cnt.cpu().numpy()
    for s in (0, 1, 7, 13, 31, 40, 55, 63):                         # 8 distinct query sets against the oracle ...
        wc, wo, _ = orc.search_proj_points(f, sets[s], 3.0, 0.8)
        for k in range(s, n_inst, n_sets):                          # ... and every instance that ran that set
            assert cnt[k] == wc and np.array_equal(owner[k], wo), (s, k)
    for k in range(n_sets, n_inst):                                 # instances of one set agree with each other
        assert cnt[k] == cnt[k - n_sets] and np.array_equal(owner[k], owner[k - n_sets]), k
    ctx.free_frames(fid); ctx.free_queries(qid)


@pytest.mark.parametrize("variant", [0, 1])
def test_c4_one_frame_against_64_candidate_keyframes(ctx, orc, variant):
    kfs = synth.keyframe_pool(seed=synth.SEED0 + 4, n_kf=65, n=N)   # common pool: ~30 % overlap with the frame each
    ids = ctx.upload_frames(kfs)
    cand = ids[1:65]; frame = np.full(64, ids[0], np.int32)
    counts, match = ctx.search_bow(variant, cand, frame, 0.75, True, stride=N)
    tot = 0
    for k in range(64):
        wc, wm, _ = orc.search_bow(variant, kfs[1 + k], kfs[0], 0.75, True)
        assert counts[k] == wc and np.array_equal(match[k], wm), (variant, k)
        tot += wc
    assert tot > 64 * 50
    ctx.free_frames(ids)


def test_c5_full_triangle_sampled_against_the_oracle(ctx, orc):
    import torch
    dev = torch.device("cuda", 0)
    n_kf = 320
    kfs = []
    for s in range(0, n_kf, 160):                                   # two trajectory chunks, as scripts/c5_allpairs.py builds its store
        kfs += synth.keyframe_pool(seed=synth.SEED0 + 5 + s, n_kf=160, n=N, window=int(N / 0.6))
    ids = ctx.upload_frames(kfs)
    tid = torch.tensor(ids.astype(np.int32), device=dev)
    pa, pb = allpairs.shard_pairs(n_kf, 0, 1, tile=32)
    assert len(pa) == n_kf * (n_kf - 1) // 2
    tpa = torch.tensor(pa, device=dev).long(); tpb = torch.tensor(pb, device=dev).long()
    rng = np.random.default_rng(5)
    near = np.nonzero(np.abs(pa - pb) <= 3)[0]                      # neighbours carry the matches: half of the sample
    sample = np.unique(np.concatenate([rng.choice(near, 100, replace=False), rng.choice(len(pa), 110, replace=False)]))
    assert len(sample) >= 200
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        for name in ("bow", "bf"):
            kept_counts = None
            for B in (16384, 5000):
                match = torch.empty((B, N), dtype=torch.int32, device=dev); counts = torch.empty(B, dtype=torch.int32, device=dev)
                all_counts = torch.empty(len(pa), dtype=torch.int32, device=dev)
                rows = {}
                for s in range(0, len(pa), B):
                    e = min(len(pa), s + B)
                    fa = tid[tpa[s:e]].contiguous(); fb = tid[tpb[s:e]].contiguous()
                    if name == "bow":
                        ctx.search_bow_dev(1, e - s, fa.data_ptr(), fb.data_ptr(), 0.75, True, N, match.data_ptr(), counts.data_ptr())
                    else:
                        ctx.bf2nn_dev(e - s, fa.data_ptr(), fb.data_ptr(), 0.75, 50, N, match.data_ptr(), None, None, counts.data_ptr())
                    all_counts[s:e] = counts[: e - s]
                    if B == 16384:
                        for k in sample[(sample >= s) & (sample < e)]:
                            rows[int(k)] = match[k - s].cpu().numpy()
                torch.cuda.synchronize()
                got = all_counts.cpu().numpy()
                if kept_counts is None:
                    kept_counts = got
                    tot = 0
                    for k in sample:
                        a, b = kfs[pa[k]], kfs[pb[k]]
                        if name == "bow":
                            wc, wm, _ = orc.search_bow(1, a, b, 0.75, True)
                        else:
                            wc, wm = orc.bf2nn(a, b, 0.75, 50)[:2]
                        assert got[k] == wc and np.array_equal(rows[int(k)], wm), (name, k, pa[k], pb[k])
                        tot += wc
                    assert tot > 5000
                else:
                    assert np.array_equal(got, kept_counts), name     # batch-size invariance of every pair's count
    finally:
        ctx.use_own_stream()
    ctx.free_frames(ids)
