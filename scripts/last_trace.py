"""Phase times of fused_kernel for one tracking-sized call (DIAG library, ORBM_WIN_TRACE): a5 and the LastFrame overload."""
import sys, os
os.environ["ORBM_DIAG_LIB"] = "1"; os.environ["ORBM_WIN_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.cuda.init()
from orbslam2_amd import synth, _lib
from orbslam2_amd.matcher import Context
ctx = Context(0)
_lib.load().orbm__window_variant(3)
f, q = synth.window_instance(700, 0, n=2000, n_queries=1500, th=7.0, stereo=True)
fid = ctx.upload_frames([f]); qid = ctx.upload_win_queries([q])
for i in range(3):
    counts, owner = ctx.search_window(0, fid, qid, 100, True, 2000)
print("LastFrame 1500 q:", counts)
fp, qp = synth.proj_instance(seed=synth.SEED0 + 9, n=2000, n_queries=5000)
fpid = ctx.upload_frames([fp]); qpid = ctx.upload_proj_queries([qp])
for i in range(3):
    counts, owner = ctx.search_proj_points(fpid, qpid, 3.0, 0.8, 2000)
print("proj 5000 q:", counts)
