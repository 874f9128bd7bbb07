"""Config 3 probe: SearchByProjection, N_INST independent (frame, 50k-query) instances per launch, distinct
query sets (permutations of one base set) so that the queries really stream from HBM."""
import sys, os, time
os.environ.setdefault("ORBM_DIAG_LIB", "1")   # the timing diagnostics live in `make -C orbslam2_amd/csrc diag` only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orbslam2_amd import synth
from orbslam2_amd.views import ProjQueries
from orbslam2_amd.matcher import Context

n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_sets = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ctx = Context(0)
f, q = synth.proj_instance(seed=synth.SEED0 + 3, n=2000, n_queries=50000)
rng = np.random.default_rng(0)
sets = []
for k in range(n_sets):
    p = rng.permutation(q.n)
    sets.append(ProjQueries(desc=q.desc[p], proj_x=q.proj_x[p], proj_y=q.proj_y[p], proj_xr=q.proj_xr[p], level=q.level[p],
                            view_cos=q.view_cos[p], flags=q.flags[p]))
fid = ctx.upload_frames([f] * 8)
qid = ctx.upload_proj_queries(sets)
dev = torch.device("cuda:0")
tfr = torch.tensor(np.resize(fid, n_inst), device=dev); tqs = torch.tensor(np.resize(qid, n_inst), device=dev)
owner = torch.empty((n_inst, 2000), dtype=torch.int32, device=dev); cnt = torch.empty(n_inst, dtype=torch.int32, device=dev)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
bytes_inst = 50000 * (32 + 20 + 1) + 2000 * 49 + 3073 * 4 + 2000 * 4
THS = [float(t) for t in os.environ.get("ORBM_TH", "1,3,5,8").split(",")]
from orbslam2_amd import _lib
_L = _lib.load()
COSTS = [int(c) for c in os.environ.get("ORBM_WIN_COST", "1").split(",")]     # 1 = exact trip-count classes, 0 = window area
for cost in COSTS:
  _L.orbm__window_cost_exact(cost)
  print("cost classes:", "exact trip count" if cost else "window area")
  for th in THS:
      ts = []
      for rep in range(4):
          e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
          e0.record()
          ctx.search_proj_points_dev(n_inst, tfr.data_ptr(), tqs.data_ptr(), th, 0.8, None, 2000, owner.data_ptr(), cnt.data_ptr())
          e1.record(); torch.cuda.synchronize()
          ts.append(e0.elapsed_time(e1))
      ms = min(ts[1:])
      print("th=%.0f  %d inst: %.3f ms  %.2f G queries/s  %.1f GB/s algorithmic (%.1f%% of 8 TB/s)  matches/inst=%d" % (
          th, n_inst, ms, n_inst * 50000 / ms / 1e6, n_inst * bytes_inst / ms / 1e6, n_inst * bytes_inst / ms / 1e6 / 80.0, int(cnt[0])))
