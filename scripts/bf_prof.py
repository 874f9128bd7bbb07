"""Runs the batched bf2nn kernel a few times (for rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orbslam2_amd import synth, _lib
from orbslam2_amd.matcher import Context
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ctx = Context(0)
_lib.load().orbm__bf2nn_variant(variant)
kfs = synth.keyframe_pool(seed=5, n_kf=64, n=2000, window=3300)
ids = ctx.upload_frames(kfs)
P = 4096
rng = np.random.default_rng(0)
pa = rng.integers(0, 64, P); pb = rng.integers(0, 64, P)
dev = torch.device("cuda:0")
fa = torch.tensor(ids[pa], dtype=torch.int32, device=dev); fb = torch.tensor(ids[pb], dtype=torch.int32, device=dev)
match = torch.empty((P, 2000), dtype=torch.int32, device=dev); counts = torch.empty(P, dtype=torch.int32, device=dev)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for _ in range(4):
    ctx.bf2nn_dev(P, fa.data_ptr(), fb.data_ptr(), 0.75, 50, 2000, match.data_ptr(), None, None, counts.data_ptr())
torch.cuda.synchronize()
print("done", int(counts.sum()))
