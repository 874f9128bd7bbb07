"""Runs the C++ parity test of the drop-in class ORB_SLAM2::ORBmatcher (tests/cpp/test_orbmatcher.cc): every
public search method, called exactly as the reference's callers do, against the CPU oracle."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_orbmatcher")


@pytest.mark.gpu
def test_cpp_orbmatcher_dropin():
    assert os.path.exists(EXE), "build it with `make -C orbslam2_amd/host` (done by __graft_entry__.build())"
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=110)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "ALL OK" in r.stdout
