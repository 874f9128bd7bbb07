"""Batched keyframe-pair matching sharded over the GPUs of one node (BASELINE config 5).

Every (keyframe i, keyframe j) pair is an independent unit, so the path shards with NO data-path collective:
each rank holds a replica of the packed keyframe store (4500 x 2000 x 32 B = 288 MB of descriptors, trivial
next to 288 GB of HBM), takes a block-cyclic share of the upper-triangular tile matrix (orbm_allpairs_shard),
matches its pairs batch by batch (orbm_allpairs_batch: match rows never leave the GPU) and hands COMPACT records
to rank 0:

    per batch and rank   one gather of (P + 1) x 8 bytes: a summary {u32 pair_id, u16 nmatches, u16 0} for every pair
                         (pair_id 0xFFFFFFFF = padding of a short last batch) + one header {u32 accepted pairs,
                         u32 list entries};
                         then the (u16 idx1, u16 idx2) lists of the pairs with nmatches >= threshold, exactly
                         `list entries` x 4 bytes, point to point (grouped send / recv = a variable-length gather).

The reference keeps a pair only if nmatches >= 20 (LoopClosing.cc:459) or 15 (Tracking.cc:1787); on a trajectory
store that is about one pair in a thousand, so the traffic is ~8 bytes per pair.  torch.distributed carries the
records (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests); communication of batch k runs on
its own stream while batch k + 1 is matched.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, List, Optional, Tuple

import numpy as np

from . import _lib

SUMMARY_DTYPE = np.dtype([("pair_id", "<u4"), ("n", "<u2"), ("pad", "<u2")])
PAD_ID = 0xFFFFFFFF


def shard_size(n_kf: int, rank: int = 0, world: int = 1, tile: int = 32) -> int:
    n = _lib.load().orbm_allpairs_shard(n_kf, tile, rank, world, 0, 0, None, None)
    if n < 0:
        raise ValueError(f"orbm_allpairs_shard({n_kf}, tile {tile}, rank {rank} of {world}) failed: {n}")
    return int(n)


def shard_pairs(n_kf: int, rank: int = 0, world: int = 1, tile: int = 32, first: int = 0,
                count: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
    """Pairs (a < b) owned by `rank`, entries [first, first + count) of its enumeration: the triangle's tile x tile
    blocks are dealt round-robin, so diagonal (half full) and off-diagonal blocks spread evenly; a block's pairs come
    from two runs of `tile` consecutive keyframes (2 x tile x 64 KB of descriptors: one XCD's L2)."""
    total = shard_size(n_kf, rank, world, tile)
    count = max(0, total - first) if count is None else max(0, min(count, total - first))
    a = np.zeros(count, np.int32); b = np.zeros(count, np.int32)
    if count:
        _lib.load().orbm_allpairs_shard(n_kf, tile, rank, world, first, count, a.ctypes.data, b.ctypes.data)
    return a, b


def pair_ids(a: np.ndarray, b: np.ndarray, n_kf: int) -> np.ndarray:
    """u32 id of pair (a, b): a * n_kf + b (n_kf <= 65535)."""
    assert n_kf <= 65535
    return (a.astype(np.uint64) * np.uint64(n_kf) + b.astype(np.uint64)).astype(np.uint32)


class PairRecords:
    """What rank 0 holds after an all-pairs run: every pair's summary and the match lists of the accepted pairs."""

    def __init__(self, n_kf: int, threshold: int):
        self.n_kf, self.threshold = n_kf, threshold
        self._summ: List[np.ndarray] = []
        self._lists: List[np.ndarray] = []
        self.bytes_received = 0          # over the wire (other ranks' records only)
        self._final = None

    def add(self, summ: np.ndarray, lists: np.ndarray, remote: bool):
        keep = summ["pair_id"] != PAD_ID
        self._summ.append(summ[keep]); self._lists.append(lists)
        if remote:
            self.bytes_received += summ.nbytes + 8 + lists.nbytes

    def _finish(self):
        if self._final is None:
            summ = np.concatenate(self._summ) if self._summ else np.zeros(0, SUMMARY_DTYPE)
            lists = np.concatenate(self._lists) if self._lists else np.zeros(0, np.uint32)
            acc = (summ["n"] >= self.threshold) & (summ["n"] > 0)
            off = np.zeros(len(summ) + 1, np.int64)
            np.cumsum(np.where(acc, summ["n"], 0), out=off[1:])
            assert off[-1] == len(lists), (off[-1], len(lists))
            self._final = (summ, lists, off, acc)
        return self._final

    @property
    def summaries(self) -> np.ndarray:
        return self._finish()[0]

    def __len__(self):
        return len(self.summaries)

    def counts(self) -> dict:
        s = self.summaries
        a, b = s["pair_id"] // self.n_kf, s["pair_id"] % self.n_kf
        return dict(zip(zip(a.tolist(), b.tolist()), s["n"].tolist()))

    def accepted(self):
        """-> (pair (a, b), idx1[], idx2[]) for every pair that reached the threshold, in arrival order."""
        summ, lists, off, acc = self._finish()
        for k in np.nonzero(acc)[0]:
            e = lists[off[k]:off[k + 1]]
            pid = int(summ["pair_id"][k])
            yield (pid // self.n_kf, pid % self.n_kf), (e & 0xFFFF).astype(np.int32), (e >> 16).astype(np.int32)

    def match_row(self, a: int, b: int, stride: int) -> Optional[np.ndarray]:
        """The pair's match row rebuilt from its list (None if the pair was not accepted or not seen)."""
        summ, lists, off, acc = self._finish()
        k = np.nonzero(summ["pair_id"] == a * self.n_kf + b)[0]
        if not len(k) or not acc[k[0]]:
            return None
        e = lists[off[k[0]]:off[k[0] + 1]]
        row = np.full(stride, -1, np.int32)
        row[e & 0xFFFF] = (e >> 16).astype(np.int32)
        return row


class RecordExchange:
    """Moves one batch's records to rank 0 (see the module docstring).  `submit` is called by every rank with its batch:
    summ = int64 tensor [P + 1] (P summaries + the header word entries << 32 | accepted), lists = int32 tensor (the first
    n_entries are meaningful).  On rank 0 `sink(rank, summ_tensor[P], lists_tensor[n])` receives every rank's batch, its own
    first.  Tensors live on `comm_device` ("cpu" for gloo, the GPU for nccl)."""

    def __init__(self, torch, dist, rank: int, world: int, batch: int, comm_device, sink: Optional[Callable], list_cap: int):
        self.torch, self.dist, self.rank, self.world, self.P = torch, dist, rank, world, batch
        self.dev, self.sink = comm_device, sink
        self.bufs = self.recv = None
        self.list_cap = list_cap
        self.overflows = 0           # batches whose lists did not fit list_cap (cut there; every rank keeps exchanging)
        if world > 1 and rank == 0:
            self.bufs = [torch.empty(batch + 1, dtype=torch.int64, device=comm_device) for _ in range(world)]
            self.recv = [torch.empty(list_cap if r else 0, dtype=torch.int32, device=comm_device) for r in range(world)]

    def submit(self, summ, lists, n_entries: int):
        torch, dist = self.torch, self.dist
        # No rank ever raises between the collectives of a batch (the others would wait in gather / irecv forever): a batch
        # whose lists exceed list_cap is CUT at list_cap on both sides -- sender and receiver compute the same length from the
        # header, which carries the true count -- and counted in `overflows`; all_pairs reports it after the last exchange.
        if n_entries > self.list_cap:
            self.overflows += 1
        n_send = min(n_entries, self.list_cap, int(lists.numel()))
        if self.world == 1:
            if self.sink:
                self.sink(0, summ[: self.P], lists[:n_send])
            return
        dist.gather(summ, self.bufs, dst=0)
        if self.rank != 0:
            if n_send:
                for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, lists[:n_send], 0)]):
                    w.wait()
            return
        hdr = torch.stack([b[self.P] for b in self.bufs]).cpu().numpy()          # one small read: every rank's header
        entries = [int(h >> 32) for h in hdr]
        ops = []
        for r in range(1, self.world):
            if entries[r] > self.list_cap:
                self.overflows += 1
                entries[r] = self.list_cap
            if entries[r]:
                ops.append(dist.P2POp(dist.irecv, self.recv[r][: entries[r]], r))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        if self.sink:
            self.sink(0, summ[: self.P], lists[:n_send])
            for r in range(1, self.world):
                self.sink(r, self.bufs[r][: self.P], self.recv[r][: entries[r]])


def all_pairs(torch, dist, n_kf: int, batch: int, threshold: int, batch_fn: Callable, rank: int = 0, world: int = 1,
              tile: int = 32, comm_device="cpu", list_cap: Optional[int] = None) -> Optional[PairRecords]:
    """Runs batch_fn over this rank's shard in batches of `batch` pairs and gathers the compact records on rank 0.

    batch_fn(pair_a, pair_b, pair_id) gets int32 / int32 / uint32 numpy arrays of length `batch` (a short last batch is
    padded with pair (0, 0), id PAD_ID) and returns (summ, lists, n_entries): an int64 tensor [batch + 1] whose entries
    are orbm_pair_summary records with the header word (entries << 32 | accepted) last, an int32 tensor of list entries
    and the number of meaningful ones -- what Context.allpairs_batch_dev leaves on the device, on `comm_device`.
    Returns PairRecords on rank 0, None elsewhere."""
    n_mine = shard_size(n_kf, rank, world, tile)
    n_max = max(shard_size(n_kf, r, world, tile) for r in range(world)) if world > 1 else n_mine
    n_batches = (n_max + batch - 1) // batch           # every rank issues the same number of exchanges
    res = PairRecords(n_kf, threshold) if rank == 0 else None

    def sink(r, summ_t, lists_t):
        s = summ_t.cpu().numpy().view(SUMMARY_DTYPE).copy()
        res.add(s, lists_t.cpu().numpy().view(np.uint32).copy(), remote=r != 0)

    ex = RecordExchange(torch, dist, rank, world, batch, comm_device, sink if rank == 0 else None,
                        list_cap if list_cap is not None else batch * 64)
    for k in range(n_batches):
        a, b = shard_pairs(n_kf, rank, world, tile, first=k * batch, count=batch)
        pid = pair_ids(a, b, n_kf)
        if len(a) < batch:
            pad = batch - len(a)
            a = np.concatenate([a, np.zeros(pad, np.int32)]); b = np.concatenate([b, np.zeros(pad, np.int32)])
            pid = np.concatenate([pid, np.full(pad, PAD_ID, np.uint32)])
        summ, lists, n_entries = batch_fn(a, b, pid)
        ex.submit(summ, lists, n_entries)
    # every exchange is done: now (and only now) an overflow may become an error, the same one on every rank
    over = ex.overflows
    if world > 1:
        t = torch.tensor([over], dtype=torch.int64, device=comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        over = int(t.item())
    if over:
        raise RuntimeError(f"{over} batch(es) produced more list entries than list_cap = {ex.list_cap}: their lists were cut "
                           f"(summaries are complete); run again with a larger list_cap")
    return res


def device_batch_fn(torch, ctx, frame_ids, kind: int, ratio: float, th: int, check_ori: bool, stride: int, threshold: int,
                    list_cap: int, comm_device):
    """batch_fn for all_pairs on a real GPU: orbm_allpairs_batch with device I/O on torch's current stream.  frame_ids =
    device int32 tensor mapping keyframe number -> frame handle.  Padded pairs match keyframe 0 against itself and are
    dropped by their id on rank 0."""
    dev = frame_ids.device
    state = {}

    def fn(a, b, pid):
        P = len(a)
        if "summ" not in state:
            state["summ"] = torch.empty(P + 1, dtype=torch.int64, device=dev)
            state["lists"] = torch.empty(list_cap, dtype=torch.int32, device=dev)
        summ, lists = state["summ"], state["lists"]
        fa = frame_ids[torch.as_tensor(a, device=dev).long()].contiguous()
        fb = frame_ids[torch.as_tensor(b, device=dev).long()].contiguous()
        tp = torch.as_tensor(pid.view(np.int32), device=dev)
        ctx.allpairs_batch_dev(kind, P, fa.data_ptr(), fb.data_ptr(), tp.data_ptr(), ratio, th, check_ori, stride, threshold,
                               summ.data_ptr(), lists.data_ptr(), list_cap, summ.data_ptr() + 8 * P)
        torch.cuda.current_stream().synchronize()
        n_entries = int(summ[P].item() >> 32)       # may exceed list_cap: RecordExchange cuts the lists and reports it at the end
        keep = min(n_entries, list_cap)
        if str(comm_device) == "cpu":
            return summ.cpu(), lists[:keep].cpu(), n_entries
        return summ.clone(), lists[:keep].clone(), n_entries

    return fn
