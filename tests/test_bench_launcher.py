"""CPU tests of bench.py's self-launch branch: `python3 bench.py --gpus N` without WORLD_SIZE starts N rank processes
itself (torch.distributed.run children, rendezvous on 127.0.0.1), relays rank 0's JSON line and a non-zero exit code if a
rank fails.  The parent never touches the GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=env, timeout=timeout)


def test_bench_launches_its_own_ranks():
    r = _run("--gpus", "2", "--launch-probe")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["ranks"] == [0, 1]


def test_bench_relays_a_failing_rank():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: the ranks would run the real benchmark")
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extra", "--no-cpu-baseline")
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
