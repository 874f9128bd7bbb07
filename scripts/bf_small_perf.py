"""Brute-force 2-NN on small batches (device lists and outputs): the default MFMA form and the vector-ALU form."""
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
match = torch.empty((4096, bench.N_FEAT), dtype=torch.int32, device=dev); counts = torch.empty(4096, dtype=torch.int32, device=dev)
for P in (1, 4, 16, 64, 256, 1024, 4096):
    fa, fb = bench.pair_list(rng, ids, P)
    tfa = torch.tensor(fa, device=dev); tfb = torch.tensor(fb, device=dev)
    for variant, name in ((0, "default"), (10, "fp4 lds"), (14, "raw nt1"), (12, "raw nt2"), (13, "raw nt4"), (11, "valu")):
        L.orbm__bf2nn_variant(variant)
        best = 1e30
        for rep in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ctx.bf2nn_dev(P, tfa.data_ptr(), tfb.data_ptr(), 0.75, 50, bench.N_FEAT, match.data_ptr(), None, None, counts.data_ptr()); e1.record()
            torch.cuda.synchronize()
            if rep: best = min(best, e0.elapsed_time(e1))
        print("%5d pairs, %-8s: %.4f ms  (%.1f G cmps/s)" % (P, name, best, P * 4e6 / best / 1e6), flush=True)
L.orbm__bf2nn_variant(0); ctx.close()
