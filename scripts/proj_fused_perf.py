"""Config 3 probe (release library): SearchByProjection, N_INST (frame, 50k-query) instances per launch, distinct query sets,
fused form against split form.  Usage: python scripts/proj_fused_perf.py [n_inst] [n_sets]; ORBM_TH=1,3,5  ORBM_FORMS=3,4"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orbslam2_amd import synth, _lib
from orbslam2_amd.views import ProjQueries
from orbslam2_amd.matcher import Context

n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n_sets = int(sys.argv[2]) if len(sys.argv) > 2 else 256
NQ = int(os.environ.get("ORBM_NQ", "50000"))       # queries per instance (prefixes of the 50k set: the early, claim-heavy part)
ctx = Context(0)
L = _lib.load()
f, q = synth.proj_instance(seed=synth.SEED0 + 3, n=2000, n_queries=50000)
OBS = float(os.environ.get("ORBM_OBS_FRAC", "1.0"))      # < 1: only this share of the frame's keypoints is ever observed by a query
if OBS < 1.0:                                             # (the other slots stay free for the whole search: no exhaustion)
    g = np.random.default_rng(123)
    src = g.integers(0, int(2000 * OBS), 50000)
    q.desc[:] = synth.observe(g, f.desc[src], 0.08)
    q.proj_x[:] = (f.x[src] + g.normal(0, 2.0, 50000)).astype(np.float32); q.proj_y[:] = (f.y[src] + g.normal(0, 2.0, 50000)).astype(np.float32)
    q.level[:] = np.clip(f.octave[src] + (g.random(50000) < 0.3), 0, 7).astype(np.int32)
rng = np.random.default_rng(0)
sets = []
for k in range(n_sets):
    p = rng.permutation(q.n)[:NQ]
    sets.append(ProjQueries(desc=q.desc[p], proj_x=q.proj_x[p], proj_y=q.proj_y[p], proj_xr=q.proj_xr[p], level=q.level[p],
                            view_cos=q.view_cos[p], flags=q.flags[p]))
fid = ctx.upload_frames([f] * 8)
qid = ctx.upload_proj_queries(sets)
del sets
dev = torch.device("cuda:0")
tfr = torch.tensor(np.resize(fid, n_inst), device=dev); tqs = torch.tensor(np.resize(qid, n_inst), device=dev)
owner = torch.empty((n_inst, 2000), dtype=torch.int32, device=dev); cnt = torch.empty(n_inst, dtype=torch.int32, device=dev)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
bytes_inst = NQ * (32 + 20 + 1) + 2000 * 49 + 3073 * 4 + 2000 * 4
THS = [float(t) for t in os.environ.get("ORBM_TH", "1,3,5").split(",")]
FORMS = [int(t) for t in os.environ.get("ORBM_FORMS", "3,4").split(",")]
ref = {}
for form in FORMS:
    L.orbm__window_variant(form)
    for th in THS:
        ts = []
        for rep in range(5):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            ctx.search_proj_points_dev(n_inst, tfr.data_ptr(), tqs.data_ptr(), th, 0.8, None, 2000, owner.data_ptr(), cnt.data_ptr())
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = min(ts[1:])
        key = (owner.cpu().numpy().copy(), cnt.cpu().numpy().copy())
        same = ""
        if th in ref:
            same = "  same as first form: %s" % (np.array_equal(ref[th][0], key[0]) and np.array_equal(ref[th][1], key[1]))
        else:
            ref[th] = key
        print("form %d th %.0f: %.3f ms  %.2f G queries/s  %.0f GB/s algorithmic = %.3f of 8 TB/s   matches/inst %.0f%s" %
              (form, th, t, n_inst * NQ / t / 1e6, n_inst * bytes_inst / t / 1e6, n_inst * bytes_inst / t / 1e6 / 8000, cnt.float().mean().item(), same), flush=True)
L.orbm__window_variant(0)
ctx.close()
