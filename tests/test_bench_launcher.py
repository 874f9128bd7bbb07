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


def test_roofline_traffic_comes_from_a_counter_pass_on_the_same_workload():
    """bench.py fills roofline.traffic from a committed rocprofv3 counter pass only when that pass was taken on the store
    size and launch size it runs itself; any other workload gets null (never a number measured on something else)."""
    import importlib.util, json
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    hit = bench.measured_traffic(bench.STORE_KF, bench.PAIRS_PER_LAUNCH)
    assert hit is not None, "profiles/r*_bf2nn_traffic.json for the default workload is missing"
    rec, src = hit
    assert rec["store_keyframes"] == bench.STORE_KF and rec["pairs_per_launch"] == bench.PAIRS_PER_LAUNCH
    assert src.startswith("profiles/") and json.load(open(os.path.join(ROOT, src)))["hbm_bytes_per_launch"] == rec["hbm_bytes_per_launch"]
    assert 1e7 < rec["hbm_bytes_per_launch"] < 5e9
    assert bench.measured_traffic(640, bench.PAIRS_PER_LAUNCH) is None
    assert bench.measured_traffic(bench.STORE_KF, 4096) is None
