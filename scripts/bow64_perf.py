"""BASELINE config 4 as stated: one frame against 64 candidate keyframes, SearchByBoW(KF, F) (variant 0) and (KF, KF) (variant 1),
with both kernel forms of the small-batch path (ORBM form knob: 0 = default, 2 = queries on lanes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from orbslam2_amd.matcher import Context
from orbslam2_amd import _lib
torch.cuda.init(); dev = torch.device("cuda:0")
ctx = Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
kfs, ids = bench.build_store(ctx, 96)
L = _lib.load()
rng = np.random.default_rng(3)
match = torch.empty((8192, bench.N_FEAT), dtype=torch.int32, device=dev); counts = torch.empty(8192, dtype=torch.int32, device=dev)
for n_pairs in [int(x) for x in os.environ.get('SIZES', '1,16,64,256,512,1024,2047').split(',')]:
    cand = ids[1 + rng.choice(len(ids) - 1, min(n_pairs, len(ids) - 1), replace=False)].astype(np.int32)
    cand = np.resize(cand, n_pairs)
    tca = torch.tensor(cand, device=dev); tcb = torch.tensor(np.full(n_pairs, ids[0], np.int32), device=dev)
    for variant in (0, 1):
        for form, wgs in ((2, 0), (0, 1), (0, 2048), (0, 4096), (0, 8192)):     # 2 = the vector-ALU kernels; 0 = matrix-core form spread over `wgs` workgroups
            L.orbm__bow_variant(form)
            L.orbm__bowm_wgs(wgs if wgs else 4096)
            best = 1e30
            for rep in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ctx.search_bow_dev(variant, n_pairs, tca.data_ptr(), tcb.data_ptr(), 0.75, True, bench.N_FEAT, match.data_ptr(), counts.data_ptr()); e1.record()
                torch.cuda.synchronize()
                if rep: best = min(best, e0.elapsed_time(e1))
            print("%4d pairs, variant %d, form %d wgs %d: %.4f ms" % (n_pairs, variant, form, wgs, best), flush=True)
L.orbm__bow_variant(0); L.orbm__bowm_wgs(4096); ctx.close()
