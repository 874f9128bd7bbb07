"""Two ranks (gloo, both on the one GPU of the test box): every rank runs the real kernels on its shard of the pair
triangle (orbm_allpairs_batch: match + compact records on the device), the records are exchanged as allpairs.py
describes (fixed-size summaries + variable-length lists) and rank 0 compares every pair with the oracle.  The N-GPU path of
bench.py differs only in the backend (RCCL) and in each rank owning its own GPU."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N_KF, N_FEAT, THRESHOLD = 20, 500, 20


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
    fn = allpairs.device_batch_fn(torch, ctx, ids, 0 if kind == "bf" else 1, 0.75, 50, True, N_FEAT, THRESHOLD,
                                  list_cap=48 * N_FEAT, comm_device="cpu")
    res = allpairs.all_pairs(torch, dist, N_KF, batch=48, threshold=THRESHOLD, batch_fn=fn, rank=rank, world=world, tile=4,
                             comm_device="cpu", list_cap=48 * N_FEAT)
    if rank == 0:
        q.put((res.counts(), [(p, i1.tolist(), i2.tolist()) for p, i1, i2 in res.accepted()], res.bytes_received,
               allpairs.shard_size(N_KF, 1, world, 4)))
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
        counts, accepted, bytes_received, n_remote = q.get(timeout=240)
    finally:
        [p.join(60) for p in procs]
        for p in procs:
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    kfs = _pool()
    assert len(counts) == N_KF * (N_KF - 1) // 2
    listed = {p: (i1, i2) for p, i1, i2 in accepted}
    total = n_listed = 0
    for (a, b), cnt in counts.items():
        if kind == "bf":
            wc, wm, _, _, _ = orc.bf2nn(kfs[a], kfs[b], 0.75, 50)
        else:
            wc, wm, _ = orc.search_bow(1, kfs[a], kfs[b], 0.75, True)
        assert cnt == wc, (a, b)
        if wc >= THRESHOLD:
            rows = np.nonzero(wm >= 0)[0]
            assert listed[(a, b)] == (rows.tolist(), wm[rows].tolist()), (a, b)
            n_listed += 1
        else:
            assert (a, b) not in listed
        total += cnt
    assert total > 500 and n_listed == len(listed) and n_listed > 5
    # what crossed between the ranks: 8 bytes per pair (+ header and last-batch padding) + 4 bytes per listed match
    assert bytes_received <= (n_remote + 48) * 8 + 8 * 8 + 4 * total
