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


def test_committed_bench_lines_follow_the_contract():
    """The lines kept under profiles/ (the trace run and the unprofiled run of the default command) carry what the driver reads:
    metric / value / unit, roofline with bound, achieved, peak, frac and a measured traffic figure, cpu_baseline with cores and
    kind -- and the headline kernel is the form that multiplies every bit of every comparison; the deferred default stands
    beside it without a fraction of the MFMA peak."""
    import json
    for name in ("r03_bench_line.json", "r03_bench_line_unprofiled.json"):
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert d["metric"].startswith("G Hamming cmps/s") and d["unit"] == "G Hamming cmps/s" and d["n_gpus"] == 1
        assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
        assert "workload" in d["config"] and "model" not in d["config"]
        r = d["roofline"]
        assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        assert r["kernel"] == "bf2nn_mfma4r_kernel<false,4,2>" and r["traffic"] and r["traffic_source"].startswith("profiles/")
        # value is whole-job throughput, the kernel's own rate can only be higher
        assert d["value"] <= r["achieved_T_cmps_per_s"] * 1e3 * 1.001
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
        f = d["roofline_deferred_form"]
        assert f["kernel"] == "bf2nn_mfma4r_kernel<false,4,4>" and "frac" not in f and f["vs_headline_form"] > 1.0
