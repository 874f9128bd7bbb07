"""A/B of bf2nn kernel variants in one process (interleaved rounds)."""
import time, sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orbslam2_amd import synth, _lib
from orbslam2_amd.matcher import Context

ctx = Context(0)
L = _lib.load()
kfs = synth.keyframe_pool(seed=5, n_kf=64, n=2000, window=3300)
ids = ctx.upload_frames(kfs)
P = 4096
rng = np.random.default_rng(0)
pa = rng.integers(0, 64, P); pb = rng.integers(0, 64, P)
dev = torch.device("cuda:0")
fa = torch.tensor(ids[pa], dtype=torch.int32, device=dev); fb = torch.tensor(ids[pb], dtype=torch.int32, device=dev)
match = torch.empty((P, 2000), dtype=torch.int32, device=dev); counts = torch.empty(P, dtype=torch.int32, device=dev)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
names = {11: "R2U4", 8: "R2U4-early-abandon", 9: "MFMA-i8", 10: "MFMA-fp4-mx", 12: "MFMA-fp4 raw rows, no LDS, 2 tiles", 13: "MFMA-fp4 raw rows, no LDS, 4 tiles"}
if os.environ.get("ORBM_BF_ALL"): names.update({1: "R4U2", 2: "R4U1", 3: "R2U2", 4: "R4U2-noupdate", 5: "R4U2-always", 6: "R4U4", 7: "R2U4-trackall"})
res = {v: [] for v in names}
for rnd in range(8):
    for v in names:
        L.orbm__bf2nn_variant(v)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.bf2nn_dev(P, fa.data_ptr(), fb.data_ptr(), 0.75, 50, 2000, match.data_ptr(), None, None, counts.data_ptr())
        e1.record(); torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1))
ref = None
for v in (11, 8, 9, 10, 12, 13):   # the early-abandon and MFMA variants must give the same matches
    L.orbm__bf2nn_variant(v)
    ctx.bf2nn_dev(P, fa.data_ptr(), fb.data_ptr(), 0.75, 50, 2000, match.data_ptr(), None, None, counts.data_ptr())
    torch.cuda.synchronize()
    cur = (match.clone(), counts.clone())
    if ref is None: ref = cur
    else: print("variant %d identical to default:" % v, bool((cur[0] == ref[0]).all() and (cur[1] == ref[1]).all()))
L.orbm__bf2nn_variant(0)
for v in names:
    t = np.array(res[v][1:])
    print("%-14s median %.3f ms  min %.3f ms  -> %.1f G cmps/s" % (names[v], np.median(t), t.min(), P * 4e6 / np.median(t) / 1e6))
