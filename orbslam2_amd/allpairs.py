"""Batched keyframe-pair matching sharded over the GPUs of one node (BASELINE config 5).

Every (keyframe i, keyframe j) pair is an independent unit, so the path shards with NO data-path collective:
each rank holds a replica of the packed keyframe store (4500 x 2000 x 32 B = 288 MB of descriptors, trivial
next to 288 GB of HBM), takes a block-cyclic share of the upper-triangular tile matrix, matches its pairs
batch by batch and hands fixed-stride match records to rank 0 with ONE gather per batch
(torch.distributed: backend "nccl" is RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
The gather of batch k overlaps the matching of batch k+1 (async work handle, double-buffered records).
"""
from __future__ import annotations

from typing import Callable, Iterator, List, Optional, Tuple

import numpy as np


def tile_pairs(n_kf: int, tile: int = 32) -> List[Tuple[int, int]]:
    """Upper-triangular tiles (ti <= tj) of the n_kf x n_kf pair matrix, row-major."""
    nt = (n_kf + tile - 1) // tile
    return [(ti, tj) for ti in range(nt) for tj in range(ti, nt)]


def shard_pairs(n_kf: int, rank: int, world: int, tile: int = 32) -> Tuple[np.ndarray, np.ndarray]:
    """Pairs (i < j) owned by `rank`: tiles are dealt round-robin so that the triangle's diagonal tiles (half
    full) and off-diagonal tiles (full) are spread evenly; both keyframes of a tile's pairs come from two
    runs of `tile` consecutive keyframes, so a tile's descriptors (2 x tile x 64 KB) sit in one XCD's L2."""
    a_all, b_all = [], []
    for k, (ti, tj) in enumerate(tile_pairs(n_kf, tile)):
        if k % world != rank:
            continue
        ia = np.arange(ti * tile, min((ti + 1) * tile, n_kf))
        ib = np.arange(tj * tile, min((tj + 1) * tile, n_kf))
        A, B = np.meshgrid(ia, ib, indexing="ij")
        keep = A < B
        a_all.append(A[keep]); b_all.append(B[keep])
    if not a_all:
        return np.zeros(0, np.int32), np.zeros(0, np.int32)
    return np.concatenate(a_all).astype(np.int32), np.concatenate(b_all).astype(np.int32)


def batches(n: int, batch: int) -> Iterator[Tuple[int, int]]:
    for s in range(0, n, batch):
        yield s, min(n, s + batch)


def pack_records(torch, counts, match, pair_a, pair_b):
    """Fixed-stride match record per pair: [a, b, count, match[0..stride)] as int32 (device or host tensor)."""
    p, stride = match.shape
    rec = torch.empty((p, stride + 3), dtype=torch.int32, device=match.device)
    rec[:, 0] = pair_a
    rec[:, 1] = pair_b
    rec[:, 2] = counts
    rec[:, 3:] = match
    return rec


def all_pairs(torch, dist, n_kf: int, stride: int, batch: int,
              match_fn: Callable, rank: int = 0, world: int = 1, tile: int = 32, device="cpu",
              keep_lists: bool = True):
    """Runs match_fn(pair_a, pair_b) -> (counts[p] int32 tensor, match[p, stride] int32 tensor) over this rank's
    pairs in batches of `batch` (the last one padded with pair (0,0) and count -1) and gathers the records on
    rank 0.  Returns on rank 0 a dict {(a, b): (count, match row or None)}; None elsewhere."""
    pa, pb = shard_pairs(n_kf, rank, world, tile)
    # all ranks must issue the same number of gathers: use the largest shard
    n_max = max(len(shard_pairs(n_kf, r, world, tile)[0]) for r in range(world))
    n_batches = (n_max + batch - 1) // batch
    results = {} if rank == 0 else None
    pending: List[Optional[tuple]] = [None, None]

    def finish(slot):
        if pending[slot] is None:
            return
        work, bufs = pending[slot]
        if work is not None:
            work.wait()
        if rank == 0:
            for t in bufs:
                arr = t.cpu().numpy()
                for row in arr:
                    if row[2] < 0:
                        continue
                    results[(int(row[0]), int(row[1]))] = (int(row[2]), row[3:].copy() if keep_lists else None)
        pending[slot] = None

    for k in range(n_batches):
        s, e = k * batch, min(len(pa), (k + 1) * batch)
        a = np.zeros(batch, np.int32); b = np.zeros(batch, np.int32)
        valid = max(0, e - s)
        if valid:
            a[:valid] = pa[s:e]; b[:valid] = pb[s:e]
        ta = torch.as_tensor(a, device=device); tb = torch.as_tensor(b, device=device)
        counts, match = match_fn(ta, tb)
        if valid < batch:
            counts = counts.clone(); counts[valid:] = -1
        rec = pack_records(torch, counts, match, ta, tb)
        slot = k & 1
        finish(slot)
        if world > 1:
            bufs = [torch.empty_like(rec) for _ in range(world)] if rank == 0 else None
            work = dist.gather(rec, bufs, dst=0, async_op=True)
            pending[slot] = (work, bufs if rank == 0 else [])
        else:
            pending[slot] = (None, [rec])
    finish(0); finish(1)
    return results
