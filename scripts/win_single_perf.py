"""Where a single windowed call spends its time through the C ABI (what ORBmatcher::SearchByProjection(Cur, Last) does per
call after the frame is resident): query upload, search with host outputs, query free; and the search alone with every
list on the device (events)."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orbslam2_amd import synth, _lib
from orbslam2_amd.matcher import Context, _addr

ctx = Context(0)
L = _lib.load()
L.orbm__window_variant(int(os.environ.get('ORBM_WIN_VARIANT', '0')))   # 0 split (collect + claims), 1 one kernel with the frame in LDS, 2 one kernel from global memory
for nq in (1000, 1500):
    f, q = synth.window_instance(11, 0, n=2000, n_queries=nq, th=7.0)
    fid = ctx.upload_frames([f])
    fr = np.ascontiguousarray(fid, np.int32)
    owner = np.full((1, 2000), -1, np.int32); counts = np.zeros(1, np.int32)
    t_up, t_se, t_fr = [], [], []
    for it in range(60):
        t0 = time.perf_counter()
        qid = ctx.upload_win_queries([q])
        t1 = time.perf_counter()
        qs = np.ascontiguousarray(qid, np.int32)
        ctx._chk(L.orbm_search_window_batch(ctx._h, 0, 1, _addr(fr), _addr(qs), 100, 1, None, 2000, _addr(owner), _addr(counts), 0))
        t2 = time.perf_counter()
        ctx.free_queries(qid)
        t3 = time.perf_counter()
        t_up.append(t1 - t0); t_se.append(t2 - t1); t_fr.append(t3 - t2)
    med = lambda v: 1e3 * float(np.median(v[10:]))
    print("n_queries %d: upload %.3f ms, search (host lists and outputs) %.3f ms, free %.3f ms; matches %d" % (nq, med(t_up), med(t_se), med(t_fr), int(counts[0])), flush=True)
    # the search alone, lists and outputs on the device
    dev = torch.device("cuda:0")
    qid = ctx.upload_win_queries([q])
    dfr = torch.tensor(fr, device=dev); dqs = torch.tensor(np.ascontiguousarray(qid, np.int32), device=dev)
    down = torch.empty((1, 2000), dtype=torch.int32, device=dev); dcnt = torch.empty(1, dtype=torch.int32, device=dev)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ts = []
    for it in range(40):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx._chk(L.orbm_search_window_batch(ctx._h, 0, 1, dfr.data_ptr(), dqs.data_ptr(), 100, 1, None, 2000, down.data_ptr(), dcnt.data_ptr(), 1))
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("   device lists: %.3f ms between events (kernels + launch gaps)" % float(np.median(ts[5:])), flush=True)
    ctx.use_own_stream()
    ctx.free_queries(qid)
    ctx.free_frames(fid)
