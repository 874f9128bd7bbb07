"""BASELINE config 5 on ONE GPU: all pairs of a resident keyframe store (default 4500 keyframes x 2000 features =
10 122 750 pairs), SearchByBoW(KF,KF) and optionally the brute-force 2-NN variant, using the pair tiling of
orbslam2_amd/allpairs.py.  Per batch only the per-pair counts stay on the device (the loop-closure acceptance
threshold, 20 matches, decides which match lists would travel); prints pairs/s.
    python scripts/c5_allpairs.py [n_kf] [bf]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orbslam2_amd import synth, allpairs
from orbslam2_amd.matcher import Context

n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 4500
do_bf = len(sys.argv) > 2 and sys.argv[2] == "bf"
N = 2000
torch.cuda.init()
ctx = Context(0)
t0 = time.time()
ids = []
CH = 250
for s in range(0, n_kf, CH):      # generate and upload in chunks (same trajectory statistics per chunk)
    kfs = synth.keyframe_pool(seed=synth.SEED0 + 5 + s, n_kf=min(CH, n_kf - s), n=N, window=int(N / 0.6))
    ids.append(ctx.upload_frames(kfs))
    print("uploaded", s + len(kfs), "keyframes, %.0f s" % (time.time() - t0), flush=True)
ids = np.concatenate(ids).astype(np.int32)
dev = torch.device("cuda:0")
tid = torch.tensor(ids, device=dev)
pa, pb = allpairs.shard_pairs(n_kf, 0, 1, tile=32)
print("pairs:", len(pa), flush=True)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
B = 16384
match = torch.empty((B, N), dtype=torch.int32, device=dev); counts = torch.empty(B, dtype=torch.int32, device=dev)
tpa = torch.tensor(pa, device=dev).long(); tpb = torch.tensor(pb, device=dev).long()
for name in (["bow"] + (["bf"] if do_bf else [])):
    accepted = torch.zeros((), dtype=torch.int64, device=dev)
    total = torch.zeros((), dtype=torch.int64, device=dev)
    torch.cuda.synchronize(); t1 = time.time()
    for s in range(0, len(pa), B):
        e = min(len(pa), s + B)
        fa = tid[tpa[s:e]].contiguous(); fb = tid[tpb[s:e]].contiguous()
        if name == "bow":
            ctx.search_bow_dev(1, e - s, fa.data_ptr(), fb.data_ptr(), 0.75, True, N, match.data_ptr(), counts.data_ptr())
        else:
            ctx.bf2nn_dev(e - s, fa.data_ptr(), fb.data_ptr(), 0.75, 50, N, match.data_ptr(), None, None, counts.data_ptr())
        accepted += (counts[: e - s] >= 20).sum(); total += counts[: e - s].sum()
    torch.cuda.synchronize(); dt = time.time() - t1
    print("%s: %d pairs in %.2f s = %.2f M pairs/s; %d pairs with >= 20 matches, %d matches in all" % (
        name, len(pa), dt, len(pa) / dt / 1e6, int(accepted), int(total)), flush=True)
