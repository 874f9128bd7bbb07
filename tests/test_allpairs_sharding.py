"""CPU tests (no GPU): pair sharding (the C entry point orbm_allpairs_shard), the exchange of compact match records with
world_size 2 on gloo, and the C ABI library loading with every symbol include/orbmatch.h declares."""
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
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _tiles_by_hand(n_kf, rank, world, tile):
    """The sharding rule restated in plain Python: upper-triangular tiles row-major, tile k to rank k % world."""
    nt = (n_kf + tile - 1) // tile
    out, k = [], 0
    for ti in range(nt):
        for tj in range(ti, nt):
            if k % world == rank:
                for a in range(ti * tile, min(n_kf, (ti + 1) * tile)):
                    for b in range(tj * tile, min(n_kf, (tj + 1) * tile)):
                        if a < b:
                            out.append((a, b))
            k += 1
    return out


@pytest.mark.parametrize("n_kf,world,tile", [(0, 1, 32), (1, 1, 32), (2, 2, 32), (97, 1, 16), (97, 2, 16), (97, 8, 16), (450, 8, 32), (31, 4, 64)])
def test_shards_partition_all_pairs(n_kf, world, tile):
    seen = set()
    sizes = []
    for r in range(world):
        a, b = allpairs.shard_pairs(n_kf, r, world, tile)
        assert (a < b).all()
        assert list(zip(a.tolist(), b.tolist())) == _tiles_by_hand(n_kf, r, world, tile)
        assert allpairs.shard_size(n_kf, r, world, tile) == len(a)
        pairs = set(zip(a.tolist(), b.tolist()))
        assert len(pairs) == len(a)
        assert not (pairs & seen)
        seen |= pairs
        sizes.append(len(a))
    assert len(seen) == n_kf * (n_kf - 1) // 2
    if n_kf >= 400:
        assert max(sizes) <= 1.15 * (sum(sizes) / world)      # balanced despite the triangle


def test_shard_windows_agree_with_the_whole_enumeration():
    a, b = allpairs.shard_pairs(211, 1, 3, 16)
    for first, count in ((0, 1), (5, 100), (len(a) - 7, 50), (len(a), 10), (1000, 4097)):
        wa, wb = allpairs.shard_pairs(211, 1, 3, 16, first=first, count=count)
        assert np.array_equal(wa, a[first:first + count]) and np.array_equal(wb, b[first:first + count])
    assert allpairs.shard_size(4500, 0, 1, 32) == 4500 * 4499 // 2
    with pytest.raises(ValueError):
        allpairs.shard_size(10, 3, 2, 32)


STRIDE = 40


def _fake_rows(a, b, pid):
    """A stand-in matcher: deterministic counts (about one pair in five reaches 20) and match rows with exactly that many entries."""
    p = len(a)
    counts = ((a.astype(np.int64) * 7 + b * 13) % 25).astype(np.int32)
    match = np.full((p, STRIDE), -1, np.int32)
    for k in range(p):
        rows = (np.arange(counts[k]) * 3 + a[k]) % STRIDE
        rows = np.unique(rows)[: counts[k]]
        counts[k] = len(rows)
        match[k, rows] = (rows * 5 + b[k]) % 60000
    return counts, match


def _fake_batch_fn(threshold):
    import packref

    def fn(a, b, pid):
        counts, match = _fake_rows(a, b, pid)
        summ, lists, (acc, ent) = packref.pack(pid, counts, match, threshold)
        t = torch.empty(len(a) + 1, dtype=torch.int64)
        t[: len(a)] = torch.from_numpy(summ.view(np.int64).copy())
        t[len(a)] = (ent << 32) | acc
        return t, torch.from_numpy(lists.view(np.int32).copy()), ent
    return fn


def _check(res, n_kf, threshold):
    assert len(res) == n_kf * (n_kf - 1) // 2
    counts = res.counts()
    n_acc = 0
    listed = {pair: (i1, i2) for pair, i1, i2 in res.accepted()}
    for a in range(n_kf):
        for b in range(a + 1, n_kf):
            c, m = _fake_rows(np.array([a]), np.array([b]), None)
            assert counts[(a, b)] == c[0]
            if c[0] >= threshold and c[0] > 0:
                n_acc += 1
                i1, i2 = listed[(a, b)]
                rows = np.nonzero(m[0] >= 0)[0]
                assert np.array_equal(i1, rows) and np.array_equal(i2, m[0][rows])
                assert np.array_equal(res.match_row(a, b, STRIDE), m[0])
            else:
                assert (a, b) not in listed and res.match_row(a, b, STRIDE) is None
    assert n_acc == len(listed) and n_acc > 10


def _worker(rank, world, port, n_kf, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = allpairs.all_pairs(torch, dist, n_kf, batch=64, threshold=20, batch_fn=_fake_batch_fn(20), rank=rank, world=world, tile=8)
    if rank == 0:
        try:
            _check(res, n_kf, 20)
            # bytes over the wire: 8 per pair (+ header) + 4 per listed match, only the other rank's share
            other = allpairs.shard_size(n_kf, 1, world, 8)
            q.put(("ok", res.bytes_received, other))
        except Exception as e:  # noqa: BLE001
            q.put(("fail", repr(e), 0))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_world2_gloo():
    n_kf, world = 45, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_kf, q)) for r in range(world)]
    [p.start() for p in procs]
    status, got, other = q.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert status == "ok", got
    assert got >= other * 8 and got < other * 8 + 4 * other * 25 + 16 * 64 * 10


def test_single_rank_needs_no_process_group():
    res = allpairs.all_pairs(torch, None, 20, batch=16, threshold=20, batch_fn=_fake_batch_fn(20))
    _check(res, 20, 20)
    assert res.bytes_received == 0


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


def test_release_library_exports_no_result_changing_knob():
    """`nm -D liborbmatch_hip.so`: the only orbm__* knobs select between forms with identical results; the timing
    diagnostics (phases of a kernel switched off) exist only in the `make diag` build."""
    import subprocess
    from orbslam2_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    knobs = {l.split()[-1] for l in out.splitlines() if " orbm__" in l}
    assert knobs <= {"orbm__bf2nn_variant", "orbm__bow_variant", "orbm__bow_split_below", "orbm__bowm_wgs", "orbm__stereo_variant",
                     "orbm__vocab_variant", "orbm__window_variant", "orbm__window_fused_rule", "orbm__window_coop"}, knobs
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"ORBM_MFMA_DIAG" not in blob


def test_no_asm_statement_writes_a_matrix_operand():
    """A matrix instruction needs two wait states after a vector write of one of its operands, and hipcc inserts them only
    behind instructions it knows: an `asm` statement that writes an FP4 fragment is a hazard the compiler cannot see (it
    lost a tenth of the matches of a SearchByBoW launch once).  In the kernels that feed MFMAs from registers
    the and-or that finishes a fragment must be C; only shifts may be `asm`."""
    import re
    for f in ("bowm.hip", "bf2nn.hip"):
        src = open(os.path.join(ROOT, "orbslam2_amd", "csrc", f)).read()
        for m in re.finditer(r'asm\s*(?:volatile)?\s*\(\s*"([^"]*)"', src):
            text = m.group(1)
            # (v_bcnt_u32_b32: the popcount chain of the vector-ALU kernel, which multiplies nothing)
            assert text == "" or text.startswith(("v_lshlrev_b64", "v_lshlrev_b32", "s_setprio", "v_bcnt_u32_b32")), (f, text)


def _overflow_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        allpairs.all_pairs(torch, dist, 45, batch=64, threshold=20, batch_fn=_fake_batch_fn(20), rank=rank, world=world, tile=8, list_cap=8)
        q.put((rank, "no error"))
    except RuntimeError as e:
        q.put((rank, "raised" if "list_cap" in str(e) else repr(e)))
    dist.barrier()
    dist.destroy_process_group()


def test_list_overflow_is_an_error_on_every_rank_not_a_hang():
    """A batch whose lists exceed list_cap must not stop one rank between two collectives: every rank finishes the
    exchange (lists cut at list_cap) and then raises the same error."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_overflow_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    got = dict(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert got == {0: "raised", 1: "raised"}, got
