"""Latency of single calls with host lists and outputs (what one SLAM thread issues), many repetitions each, so that a
kernel / memory-copy trace of the run (rocprofv3 --kernel-trace --memory-copy-trace --stats) can be set beside the call time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
from orbslam2_amd import synth
from orbslam2_amd.matcher import Context
ctx = Context(0)
from orbslam2_amd import _lib
_lib.load().orbm__window_variant(int(os.environ.get("ORBM_WIN_VARIANT", "0")))
REPS = int(os.environ.get("ORBM_REPS", "300"))

def timeit(name, fn):
    for _ in range(5):
        fn()
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(REPS):
        fn()
    dt = (time.perf_counter() - t0) / REPS
    print("%-46s %.1f us per call" % (name, dt * 1e6), flush=True)

f, q = synth.window_instance(700, 0, n=2000, n_queries=1500, th=7.0, stereo=True)
fid = ctx.upload_frames([f])
timeit("search_window_once LastFrame 1500 q th 7", lambda: ctx.search_window_once(0, fid[0], q, 100, True, 2000))
qid = ctx.upload_win_queries([q])
timeit("search_window (resident query set)", lambda: ctx.search_window(0, fid, qid, 100, True, 2000))
fp, qp = synth.proj_instance(seed=synth.SEED0 + 9, n=2000, n_queries=5000)
fpid = ctx.upload_frames([fp])
timeit("search_proj_points_once 5000 q th 3", lambda: ctx.search_proj_points_once(fpid[0], qp, 3.0, 0.8, 2000))
f1, f2, prev = synth.init_pair()
ids = ctx.upload_frames([f1, f2])
timeit("search_init 1000 x 1000 window 100", lambda: ctx.search_init(ids[0], ids[1], prev, 100, 0.9, True))
kfs = synth.keyframe_pool(seed=synth.SEED0 + 5, n_kf=2, n=2000, window=int(2000 / 0.6))
kid = ctx.upload_frames(kfs)
timeit("search_bow one pair (KF,KF)", lambda: ctx.search_bow(1, kid[:1], kid[1:2], 0.75, True, 2000))
ctx.close()
