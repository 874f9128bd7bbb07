"""Descriptor stage of ComputeStereoMatches: one rectified stereo pair alone (the per-frame case) and 256 pairs per call."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
from orbslam2_amd import synth
from orbslam2_amd.matcher import Context
from oracle import oracle_ffi as orc
ctx = Context(0)
rng = np.random.default_rng(1)
n = 2000
lm = synth.random_descriptors(rng, n)
xl = rng.uniform(16, 624, n).astype(np.float32); yl = rng.uniform(16, 464, n).astype(np.float32)
left = synth.make_frame(rng, lm, np.arange(n), x=xl, y=yl, with_fv=False)
perm = rng.permutation(n)
right = synth.make_frame(rng, lm, perm, x=(xl - rng.uniform(0, 50, n).astype(np.float32))[perm], y=(yl + rng.normal(0, 1.0, n).astype(np.float32))[perm],
                         octave=left.octave[perm], with_fv=False)
ids = ctx.upload_frames([left, right])
for m in (1, 256):
    fl = np.full(m, ids[0], np.int32); fr = np.full(m, ids[1], np.int32); md = np.full(m, 60.0, np.float32)
    ts = []
    for rep in range(6):
        ctx.synchronize(); t0 = time.perf_counter()
        idx, dist = ctx.stereo_match(fl, fr, md, stride=n)
        ts.append(time.perf_counter() - t0)
    print("%d pair(s): %.3f ms per call (host lists and outputs), %d of %d left keypoints matched" % (m, min(ts[1:]) * 1e3, int((idx[0] >= 0).sum()), n))
t0 = time.perf_counter(); wi, wd, _ = orc.stereo_match(left, right, 480, 60.0); t1 = time.perf_counter()
assert np.array_equal(wi, idx[0]) and np.array_equal(wd, dist[0])
print("CPU restatement (row table, one core): %.3f ms" % ((t1 - t0) * 1e3))
