"""Single-instance windowed calls of tracking size (the lane-parallel walk of the split form, the one-call entry points,
SearchForInitialization's cooperative search) against the oracle over random shapes.  Test infrastructure, not the product
path.  Usage: python scripts/fuzz_single.py [iterations] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
torch.cuda.init()
from oracle import oracle_ffi as orc
from orbslam2_amd import synth
from orbslam2_amd.matcher import Context

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = Context(0)
t0 = time.time()
for it in range(iters):
    seed = int(rng.integers(1, 1 << 30))
    n = int(rng.choice([60, 300, 900, 1500, 2000, 2300, 3500]))
    kind = int(rng.integers(0, 5))
    if kind < 3:                                           # a11 overloads
        mode = kind
        nq = int(rng.choice([1, 63, 64, 65, 500, 1500, 4000, 8192, 8300]))
        th = float(rng.choice([3.0, 7.0, 15.0, 30.0]))
        f, q = synth.window_instance(seed, mode, n=n, n_queries=nq, th=th, stereo=(mode == 0 and rng.random() < 0.5))
        dth = int(rng.choice([50, 64, 100]))
        ori = bool(mode < 2 and rng.random() < 0.7)
        fid = ctx.upload_frames([f])
        cnt, owner = ctx.search_window_once(mode, fid[0], q, dth, ori, n)
        wc, wo, _ = orc.search_window(mode, f, q, dth, ori)
        assert cnt == wc and np.array_equal(owner[:f.n], wo), ("window", it, seed, mode, n, nq, th, dth, ori)
        qid = ctx.upload_win_queries([q])
        c2, o2 = ctx.search_window(mode, fid, qid, dth, ori, n)
        assert c2[0] == wc and np.array_equal(o2[0, :f.n], wo), ("window batch", it, seed)
        ctx.free_queries(qid); ctx.free_frames(fid)
    elif kind == 3:                                        # a6
        nq = int(rng.choice([1, 200, 1000, 5000, 8192, 9000]))
        th = float(rng.choice([1.0, 3.0, 5.0, 8.0]))
        f, q = synth.proj_instance(seed=seed, n=n, n_queries=nq, claimed_frac=float(rng.random() * 0.5), obs_frac=float(rng.random()), stereo=bool(rng.random() < 0.5))
        fid = ctx.upload_frames([f])
        cnt, owner = ctx.search_proj_points_once(fid[0], q, th, 0.8, n)
        wc, wo, _ = orc.search_proj_points(f, q, th, 0.8)
        assert cnt == wc and np.array_equal(owner[:f.n], wo), ("proj", it, seed, n, nq, th)
        ctx.free_frames(fid)
    else:                                                  # a5
        n1 = int(rng.choice([100, 500, 1000, 2000]))
        f1, f2, prev = synth.init_pair(seed=seed, n=n1)
        win = int(rng.choice([30, 100, 150, 400]))
        ids = ctx.upload_frames([f1, f2])
        cnt, m12, p2 = ctx.search_init(ids[0], ids[1], prev, win, 0.9, True)
        wc, wm, wp, _ = orc.search_init(f1, f2, prev, win, 0.9, True)
        assert cnt == wc and np.array_equal(m12, wm) and np.array_equal(p2, wp), ("init", it, seed, n1, win)
        ctx.free_frames(ids)
    if it % 20 == 19:
        print("%d ok (%.0f s)" % (it + 1, time.time() - t0), flush=True)
ctx.close()
print("ALL %d OK" % iters)
