"""Runs single windowed instances of several shapes, 20 launches each, in a fixed order: read the kernel trace of this script
(rocprofv3 --kernel-trace) with scripts/win_single_trace_read.py to see collect / claims kernel time per shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orbslam2_amd import synth, _lib
from orbslam2_amd.matcher import Context

ctx = Context(0)
L = _lib.load()
dev = torch.device("cuda:0")
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
SHAPES = [(mode, nq, th, ori) for mode in (0, 2) for nq in (10, 100, 1000) for th in (3.0, 7.0) for ori in (0, 1)]
for mode, nq, th, ori in SHAPES:
    f, q = synth.window_instance(11, mode, n=2000, n_queries=nq, th=th)
    fid = ctx.upload_frames([f]); qid = ctx.upload_win_queries([q])
    dfr = torch.tensor(np.ascontiguousarray(fid, np.int32), device=dev); dqs = torch.tensor(np.ascontiguousarray(qid, np.int32), device=dev)
    down = torch.empty((1, 2000), dtype=torch.int32, device=dev); dcnt = torch.empty(1, dtype=torch.int32, device=dev)
    for it in range(20):
        ctx._chk(L.orbm_search_window_batch(ctx._h, mode, 1, dfr.data_ptr(), dqs.data_ptr(), 100, ori, None, 2000, down.data_ptr(), dcnt.data_ptr(), 1))
    torch.cuda.synchronize()
    print("shape", mode, nq, th, ori, "matches", int(dcnt.item()), flush=True)
    ctx.free_queries(qid); ctx.free_frames(fid)
