import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.cuda.init()
import bench
from orbslam2_amd.matcher import Context
from orbslam2_amd import _lib
ctx = Context(0); L = _lib.load()
kfs, ids = bench.build_store(ctx, 8)
for form in (0, 1, 2, 0):
    L.orbm__bow_variant(form)
    for variant in (0, 1):
        ts = []
        for rep in range(30):
            t0 = time.perf_counter(); c, m = ctx.search_bow(variant, ids[[0]], ids[[1]], 0.75, True, stride=2000); ts.append(time.perf_counter() - t0)
        print("form %d variant %d single pair host path: %.4f ms (min) %.4f (median)" % (form, variant, min(ts[5:]) * 1e3, np.median(ts[5:]) * 1e3), flush=True)
L.orbm__bow_variant(0)
from orbslam2_amd import synth
c = synth.triang_pair(77, n=2000)
tid = ctx.upload_frames(list(c[:2]))
for form in (0, 2, 1):
    L.orbm__bow_variant(form)
    for P in (1, 20):
        ts = []
        fa, fb = np.full(P, tid[0]), np.full(P, tid[1])
        f12 = np.tile(c[2], (P, 1)); ep = np.tile(c[3], (P, 1))
        for rep in range(30):
            t0 = time.perf_counter(); cnt, m = ctx.search_triang(fa, fb, f12, ep, False, True, stride=2000); ts.append(time.perf_counter() - t0)
        print("form %d SearchForTriangulation %d pair(s), host path: %.4f ms (min), %d matches" % (form, P, min(ts[5:]) * 1e3, cnt[0]), flush=True)
L.orbm__bow_variant(0)
