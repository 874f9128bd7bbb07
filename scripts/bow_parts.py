"""Where the time of the matrix-core SearchByBoW goes (DIAG library: parts switched off, WRONG results): ORBM_DIAG_LIB=1."""
import sys, os
os.environ["ORBM_DIAG_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from orbslam2_amd.matcher import Context
from orbslam2_amd import _lib

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
L = _lib.load()
for bits, what in ((0, "everything"), (4, "matches not applied"), (1, "records not settled"), (2, "no records made")):
    L.orbm__bowm_diag(bits)
    for name, (fa, fb) in mixes.items():
        tfa = torch.tensor(fa, device=dev); tfb = torch.tensor(fb, device=dev)
        best = 1e30
        for rep in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ctx.search_bow_dev(1, P, tfa.data_ptr(), tfb.data_ptr(), 0.75, True, bench.N_FEAT, match.data_ptr(), counts.data_ptr())
            e1.record(); torch.cuda.synchronize()
            if rep:
                best = min(best, e0.elapsed_time(e1))
        print("%-22s %-10s %.3f ms  %.2f M pairs/s" % (what, name, best, P / best / 1e3), flush=True)
L.orbm__bowm_diag(0)
import ctypes
buf = (ctypes.c_ulonglong * 512)()
for name, (fa, fb) in mixes.items():
    tfa = torch.tensor(fa, device=dev); tfb = torch.tensor(fb, device=dev)
    for rep in range(2):
        ctx.search_bow_dev(1, P, tfa.data_ptr(), tfb.data_ptr(), 0.75, True, bench.N_FEAT, match.data_ptr(), counts.data_ptr())
    torch.cuda.synchronize()
    L.orbm__bowm_trace(buf)
    t = np.array(buf[:], dtype=np.float64).reshape(64, 8) / 100.0      # 100 MHz wall clock -> us
    d = np.diff(t[:, :6], axis=1)
    print("%-10s wave 0 of 64 sampled workgroups, us: set-up %.2f | node look-up %.2f | tiles %.2f | settle %.2f | end %.2f | life %.2f; kernel span %.1f"
          % (name, *d.mean(axis=0), (t[:, 5] - t[:, 0]).mean(), t[:, 5].max() - t[:, 0].min()), flush=True)
ctx.close()
