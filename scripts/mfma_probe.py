"""Timing of the MFMA brute-force kernel alone (variant 9); ORBM_MFMA_DIAG=1 drops the selection epilogue, =2 stages the
candidate tile only once (both give wrong results: attribution only)."""
import sys, os
os.environ.setdefault("ORBM_DIAG_LIB", "1")   # the timing diagnostics live in `make -C orbslam2_amd/csrc diag` only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orbslam2_amd import synth, _lib
from orbslam2_amd.matcher import Context
torch.cuda.init()
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
L.orbm__bf2nn_variant(int(os.environ.get('ORBM_VARIANT', '9')))
ts = []
for rnd in range(6):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    ctx.bf2nn_dev(P, fa.data_ptr(), fb.data_ptr(), 0.75, 50, 2000, match.data_ptr(), None, None, counts.data_ptr())
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
t = np.array(ts[1:])
print("variant=" + os.environ.get("ORBM_VARIANT", "9") + " diag=%s  median %.3f ms  min %.3f ms  -> %.1f G cmps/s, matches %d" % (os.environ.get("ORBM_MFMA_DIAG", "0"), np.median(t), t.min(), P * 4e6 / np.median(t) / 1e6, int(counts.sum())))
