"""Two ranks (gloo, both on the one GPU of the test box): every rank runs the real kernels on its shard of the pair
triangle, the match records are gathered on rank 0 and compared with the oracle pair by pair.  The N-GPU path of
bench.py / allpairs.py differs only in the backend (RCCL) and in each rank owning its own GPU."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N_KF, N_FEAT = 20, 500


def _pool():
    from orbslam2_amd import synth
    return synth.keyframe_pool(seed=4242, n_kf=N_KF, n=N_FEAT, window=800)


def _worker(rank, world, port, kind, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.init()
    from orbslam2_amd import allpairs
    from orbslam2_amd.matcher import Context
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = Context(0)
    ids = torch.tensor(ctx.upload_frames(_pool()).astype(np.int32), device="cuda:0")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    def match_fn(ta, tb):           # pair lists arrive on the host (gloo), the kernels run on device-resident lists
        fa = ids[ta.long().cuda()].contiguous(); fb = ids[tb.long().cuda()].contiguous()
        p = fa.numel()
        match = torch.empty((p, N_FEAT), dtype=torch.int32, device="cuda:0")
        counts = torch.empty(p, dtype=torch.int32, device="cuda:0")
        if kind == "bf":
            ctx.bf2nn_dev(p, fa.data_ptr(), fb.data_ptr(), 0.75, 50, N_FEAT, match.data_ptr(), None, None, counts.data_ptr())
        else:
            ctx.search_bow_dev(1, p, fa.data_ptr(), fb.data_ptr(), 0.75, True, N_FEAT, match.data_ptr(), counts.data_ptr())
        torch.cuda.synchronize()
        return counts.cpu(), match.cpu()

    res = allpairs.all_pairs(torch, dist, N_KF, stride=N_FEAT, batch=48, match_fn=match_fn, rank=rank, world=world, tile=4)
    if rank == 0:
        q.put({k: (v[0], v[1].tolist()) for k, v in res.items()})
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


@pytest.mark.parametrize("kind", ["bf", "bow"])
def test_two_ranks_match_their_shards_and_rank0_gets_every_pair(orc, kind):
    world = 2
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = 29700 + os.getpid() % 1500 + (0 if kind == "bf" else 1)
    procs = [mpc.Process(target=_worker, args=(r, world, port, kind, q)) for r in range(world)]
    [p.start() for p in procs]
    try:
        res = q.get(timeout=240)
    finally:
        [p.join(60) for p in procs]
        for p in procs:
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    kfs = _pool()
    assert len(res) == N_KF * (N_KF - 1) // 2
    total = 0
    for (a, b), (cnt, row) in res.items():
        if kind == "bf":
            wc, wm, _, _, _ = orc.bf2nn(kfs[a], kfs[b], 0.75, 50)
        else:
            wc, wm, _ = orc.search_bow(1, kfs[a], kfs[b], 0.75, True)
        assert cnt == wc and row[:kfs[a].n] == wm.tolist(), (a, b)
        total += cnt
    assert total > 500
