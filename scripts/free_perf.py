"""Claim-free windowed best-1 (SearchBySim3 / Fuse form, orbm_search_window_free_batch): one instance (the per-call case)
and 256 instances per call, 2000 queries into a 2000-keypoint frame, th = 3 (Fuse's default radius factor)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
from orbslam2_amd import synth
from orbslam2_amd.matcher import Context
from oracle import oracle_ffi as orc
ctx = Context(0)
insts = [synth.window_instance(900 + k, 2, n=2000, n_queries=2000, th=3.0) for k in range(16)]
fid = ctx.upload_frames([f for f, _ in insts]); qid = ctx.upload_win_queries([q for _, q in insts])
for m in (1, 256):
    fr = np.resize(fid, m); qs = np.resize(qid, m)
    ts = []
    for rep in range(6):
        ctx.synchronize(); t0 = time.perf_counter()
        best = ctx.search_window_free(fr, qs, 50, False, 2000)
        ts.append(time.perf_counter() - t0)
    print("%d instance(s): %.3f ms per call (host lists and outputs), %d of 2000 queries found a keypoint" % (m, min(ts[1:]) * 1e3, int((best[0] >= 0).sum())))
t0 = time.perf_counter(); w = orc.search_window_free(insts[0][0], insts[0][1], 50, False)[0]; t1 = time.perf_counter()
assert np.array_equal(w, best[0])
print("CPU restatement, one instance on one core: %.3f ms" % ((t1 - t0) * 1e3))
