"""Per-frame tracking calls as single instances (what one Tracking thread issues): SearchByProjection(CurrentFrame,
LastFrame) (mode 0, th = 7 and 15), the keyframe / Sim3 overloads, SearchForInitialization.  Host lists and outputs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
from orbslam2_amd import synth
from orbslam2_amd.matcher import Context
from oracle import oracle_ffi as orc
ctx = Context(0)
from orbslam2_amd import _lib
_lib.load().orbm__window_variant(int(os.environ.get('ORBM_WIN_VARIANT', '0')))
for mode, dth, th, nq in ((0, 100, 7.0, 1500), (0, 100, 15.0, 1500), (1, 100, 7.0, 1500), (2, 50, 10.0, 1500)):
    f, q = synth.window_instance(700 + mode, mode, n=2000, n_queries=nq, th=th, stereo=(mode == 0))
    fid = ctx.upload_frames([f]); qid = ctx.upload_win_queries([q])
    ts = []
    for rep in range(8):
        ctx.synchronize(); t0 = time.perf_counter()
        counts, owner = ctx.search_window(mode, fid, qid, dth, True, 2000)
        ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); c, o, _ = orc.search_window(mode, f, q, dth, True); t1 = time.perf_counter()
    assert c == counts[0] and np.array_equal(o, owner[0, :f.n])
    print("mode %d th %.0f, %d queries: %.3f ms per call (%d matches); CPU restatement %.3f ms" % (mode, th, nq, min(ts[2:]) * 1e3, c, (t1 - t0) * 1e3))
    ctx.free_frames(fid); ctx.free_queries(qid)
f1, f2, prev = synth.init_pair(seed=5, n=1000)
ids = ctx.upload_frames([f1, f2])
ts = []
for rep in range(8):
    ctx.synchronize(); t0 = time.perf_counter()
    cnt, m12, p2 = ctx.search_init(ids[0], ids[1], prev, 100, 0.9, True)
    ts.append(time.perf_counter() - t0)
t0 = time.perf_counter(); c, m, p, _ = orc.search_init(f1, f2, prev, 100, 0.9, True); t1 = time.perf_counter()
assert c == cnt and np.array_equal(m, m12)
print("SearchForInitialization 1000 x 1000, window 100: %.3f ms per call (%d matches); CPU restatement %.3f ms" % (min(ts[2:]) * 1e3, c, (t1 - t0) * 1e3))
