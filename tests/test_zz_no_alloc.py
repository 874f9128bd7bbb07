"""The allocation-free entry point orbm_search_bow_batch_noalloc against orbm_search_bow_batch (device I/O) and the oracle:
two calls in a row, the second with a smaller input (the static scratch buffer keeps the first call's contents)."""
import numpy as np
import pytest

from orbslam2_amd import synth
from orbslam2_amd.matcher import OrbmError

pytestmark = pytest.mark.gpu


def test_noalloc_bow_matches_the_allocating_entry_point(ctx, orc):
    import torch
    dev = torch.device("cuda", 0)
    n = 900
    kfs = synth.keyframe_pool(seed=synth.SEED0 + 71, n_kf=12, n=n, window=1500)
    ids = ctx.upload_frames(kfs)
    rng = np.random.default_rng(3)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        for variant in (0, 1):
            for n_pairs in (700, 90):                                  # second call smaller than the first
                ia = rng.integers(0, 12, n_pairs); ib = (ia + rng.integers(1, 4, n_pairs)) % 12
                fa = torch.tensor(ids[ia].astype(np.int32), device=dev); fb = torch.tensor(ids[ib].astype(np.int32), device=dev)
                want_m = torch.full((n_pairs, n), -9, dtype=torch.int32, device=dev); want_c = torch.full((n_pairs,), -9, dtype=torch.int32, device=dev)
                got_m = torch.full((n_pairs, n), -8, dtype=torch.int32, device=dev); got_c = torch.full((n_pairs,), -8, dtype=torch.int32, device=dev)
                ctx.search_bow_dev(variant, n_pairs, fa.data_ptr(), fb.data_ptr(), 0.75, True, n, want_m.data_ptr(), want_c.data_ptr())
                ctx.search_bow_noalloc(variant, n_pairs, fa.data_ptr(), fb.data_ptr(), 0.75, True, n, got_m.data_ptr(), got_c.data_ptr())
                torch.cuda.synchronize()
                assert torch.equal(got_c, want_c) and torch.equal(got_m, want_m), (variant, n_pairs)
                gc = got_c.cpu().numpy(); gm = got_m.cpu().numpy()
                for k in range(0, n_pairs, 37):                        # and a sample against the oracle itself
                    wc, wm, _ = orc.search_bow(variant, kfs[ia[k]], kfs[ib[k]], 0.75, True)
                    assert gc[k] == wc and np.array_equal(gm[k], wm), (variant, n_pairs, k)
                assert gc.sum() > 0
        with pytest.raises(OrbmError, match="exceed the static scratch"):
            ctx.search_bow_noalloc(1, 2049, 0, 0, 0.75, True, n, 0, 0)
    finally:
        ctx.use_own_stream()
    ctx.free_frames(ids)
