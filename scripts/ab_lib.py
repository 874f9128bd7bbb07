"""Run a script or pytest selection against another build of the library (A/B of kernel changes): python scripts/ab_lib.py <dir under orbslam2_amd> <script.py | -k expr>"""
import os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orbslam2_amd._lib as L
if sys.argv[1] != "default":
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), sys.argv[1], "liborbmatch_hip.so")
print("library:", L.LIB_PATH, flush=True)
if sys.argv[2] == "-k":
    import pytest
    sys.exit(pytest.main(["tests", "-m", "gpu", "-x", "-q", "--timeout", "120", "--timeout-method", "thread", "-k", sys.argv[3]]))
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
