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
