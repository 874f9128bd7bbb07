import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle_ffi
    oracle_ffi.lib()
    return oracle_ffi


@pytest.fixture(scope="session")
def ctx():
    # tests that hand torch device tensors to the library need torch's HIP runtime initialised first: once
    # liborbmatch_hip.so has brought up the system runtime, torch's own initialisation no longer finds the GPU
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    from orbslam2_amd.matcher import Context
    c = Context(0)
    yield c
    c.close()
