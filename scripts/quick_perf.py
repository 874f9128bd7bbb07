"""First-light perf probe (not the bench): popcount peak + batched bf2nn throughput."""
import time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orbslam2_amd import synth
from orbslam2_amd.matcher import Context

ctx = Context(0)
print(ctx.device_info())
for it in (2048, 8192):
    print("popcount peak lane-ops/s: %.3e" % ctx.popcount_peak(it))
kfs = synth.keyframe_pool(seed=5, n_kf=64, n=2000, window=3300)
ids = ctx.upload_frames(kfs)
P = 4096
rng = np.random.default_rng(0)
pa = rng.integers(0, 64, P); pb = rng.integers(0, 64, P)
dev = torch.device("cuda:0")
fa = torch.tensor(ids[pa], dtype=torch.int32, device=dev); fb = torch.tensor(ids[pb], dtype=torch.int32, device=dev)
match = torch.empty((P, 2000), dtype=torch.int32, device=dev); counts = torch.empty(P, dtype=torch.int32, device=dev)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for rep in range(3):
    torch.cuda.synchronize(); t = time.time()
    ctx.bf2nn_dev(P, fa.data_ptr(), fb.data_ptr(), 0.75, 50, 2000, match.data_ptr(), None, None, counts.data_ptr())
    torch.cuda.synchronize(); dt = time.time() - t
    print("bf2nn %d pairs: %.3f ms  %.1f G cmps/s  %.0f pairs/s" % (P, dt * 1e3, P * 4e6 / dt / 1e9, P / dt))
print(counts[:8].tolist())
