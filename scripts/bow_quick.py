"""SearchByBoW(KF,KF), matrix-core form only: bench mix / neighbours / far, 8192 pairs per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from orbslam2_amd.matcher import Context

torch.cuda.init()
dev = torch.device("cuda:0")
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
kfs, ids = bench.build_store(ctx, 96)
P = 8192
rng = np.random.default_rng(3)
a = rng.integers(0, len(ids), P)
mixes = {"bench mix": bench.pair_list(np.random.default_rng(3), ids, P),
         "neighbours": (ids[a].astype(np.int32), ids[(a + rng.integers(1, 4, P)) % len(ids)].astype(np.int32)),
         "far": (ids[a].astype(np.int32), ids[(a + rng.integers(30, 60, P)) % len(ids)].astype(np.int32))}
match = torch.empty((P, bench.N_FEAT), dtype=torch.int32, device=dev); counts = torch.empty(P, dtype=torch.int32, device=dev)
for variant in (1, 0):
    for name, (fa, fb) in mixes.items():
        tfa = torch.tensor(fa, device=dev); tfb = torch.tensor(fb, device=dev)
        best = 1e30
        for rep in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ctx.search_bow_dev(variant, P, tfa.data_ptr(), tfb.data_ptr(), 0.75, True, bench.N_FEAT, match.data_ptr(), counts.data_ptr())
            e1.record(); torch.cuda.synchronize()
            if rep:
                best = min(best, e0.elapsed_time(e1))
        print("variant %d %-10s %.3f ms  %.2f M pairs/s  mean matches %.1f" % (variant, name, best, P / best / 1e3, counts.float().mean().item()), flush=True)
ctx.close()
