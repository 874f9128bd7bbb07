"""CPU tests (no GPU): pair sharding and the gather of match records with world_size 2 on gloo, and the C ABI
library loading with every symbol include/orbmatch.h declares."""
import os
import re
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from orbslam2_amd import allpairs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n_kf,world,tile", [(1, 1, 32), (2, 2, 32), (97, 1, 16), (97, 2, 16), (97, 8, 16), (450, 8, 32), (31, 4, 64)])
def test_shards_partition_all_pairs(n_kf, world, tile):
    seen = set()
    sizes = []
    for r in range(world):
        a, b = allpairs.shard_pairs(n_kf, r, world, tile)
        assert (a < b).all()
        pairs = set(zip(a.tolist(), b.tolist()))
        assert len(pairs) == len(a)
        assert not (pairs & seen)
        seen |= pairs
        sizes.append(len(a))
    assert len(seen) == n_kf * (n_kf - 1) // 2
    if n_kf >= 400:
        assert max(sizes) <= 1.15 * (sum(sizes) / world)      # balanced despite the triangle


def _fake_match(stride):
    def fn(ta, tb):
        counts = ((ta * 7 + tb * 13) % 50).to(torch.int32)
        match = (ta[:, None] * 1000 + tb[:, None] + torch.arange(stride, dtype=torch.int32)[None, :]).to(torch.int32)
        return counts, match
    return fn


def _worker(rank, world, port, n_kf, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = allpairs.all_pairs(torch, dist, n_kf, stride=5, batch=64, match_fn=_fake_match(5), rank=rank, world=world, tile=8)
    if rank == 0:
        q.put({k: (v[0], v[1].tolist()) for k, v in res.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_gather_world2_gloo():
    n_kf, world = 45, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_kf, q)) for r in range(world)]
    [p.start() for p in procs]
    res = q.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert len(res) == n_kf * (n_kf - 1) // 2
    for (a, b), (cnt, row) in res.items():
        assert cnt == (a * 7 + b * 13) % 50
        assert row == [a * 1000 + b + k for k in range(5)]


def test_single_rank_needs_no_process_group():
    res = allpairs.all_pairs(torch, None, 20, stride=3, batch=16, match_fn=_fake_match(3))
    assert len(res) == 190 and res[(3, 7)][0] == (3 * 7 + 7 * 13) % 50


def test_c_abi_exports_every_declared_symbol():
    """The library must load on a box without a GPU and export everything include/orbmatch.h declares."""
    from orbslam2_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "orbmatch.h")).read()
    declared = set(re.findall(r"\b(orbm_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"orbm_frame_view", "orbm_proj_queries", "orbm_win_queries", "orbm_ctx"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    L = _lib.load()
    for name in declared:
        assert hasattr(L, name)


def test_no_gpu_is_a_loud_error():
    """Without a device the context cannot be created: the product path raises instead of falling back."""
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from orbslam2_amd.matcher import Context, OrbmError
    with pytest.raises(OrbmError):
        Context(0)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "orbslam2_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_ffi" not in src and "liborb_oracle" not in src and "orc_" not in src, f
