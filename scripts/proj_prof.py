"""One-threshold version of proj_perf.py for rocprofv3 counter passes: python3 scripts/proj_prof.py TH [N_INST]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orbslam2_amd import synth
from orbslam2_amd.views import ProjQueries
from orbslam2_amd.matcher import Context

th = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
n_inst = int(sys.argv[2]) if len(sys.argv) > 2 else 512
ctx = Context(0)
f, q = synth.proj_instance(seed=synth.SEED0 + 3, n=2000, n_queries=50000)
rng = np.random.default_rng(0)
sets = []
for k in range(64):
    p = rng.permutation(q.n)
    sets.append(ProjQueries(desc=q.desc[p], proj_x=q.proj_x[p], proj_y=q.proj_y[p], proj_xr=q.proj_xr[p], level=q.level[p],
                            view_cos=q.view_cos[p], flags=q.flags[p]))
fid = ctx.upload_frames([f] * 8)
qid = ctx.upload_proj_queries(sets)
dev = torch.device("cuda:0")
tfr = torch.tensor(np.resize(fid, n_inst), device=dev); tqs = torch.tensor(np.resize(qid, n_inst), device=dev)
owner = torch.empty((n_inst, 2000), dtype=torch.int32, device=dev); cnt = torch.empty(n_inst, dtype=torch.int32, device=dev)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for rep in range(3):
    ctx.search_proj_points_dev(n_inst, tfr.data_ptr(), tqs.data_ptr(), th, 0.8, None, 2000, owner.data_ptr(), cnt.data_ptr())
torch.cuda.synchronize()
print("done", int(cnt[0]))
