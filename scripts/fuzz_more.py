"""Runs the crowded-instance fuzz tests of tests/test_gpu_search.py over many seeds in ONE process (the tests compare the
GPU results with the oracle; this is test infrastructure, not the product path).  Usage: python scripts/fuzz_more.py [seeds] [first]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.cuda.init()
from oracle import oracle_ffi as orc
from orbslam2_amd.matcher import Context
import test_gpu_search as T

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
FORMS = tuple(int(x) for x in os.environ.get('ORBM_FUZZ_FORMS', '1,2,5').split(','))
WIN_FORMS = tuple(int(x) for x in os.environ.get('ORBM_FUZZ_WIN_FORMS', '3,4').split(','))
ONLY_BOW = os.environ.get('ORBM_FUZZ_ONLY_BOW') == '1'
orc.lib()
ctx = Context(0)
t0 = time.time()
for s in range(first, first + n_seeds):
    os.environ["ORBM_FUZZ_SEED"] = str(1000 * s)
    if not ONLY_BOW:
        for wv in WIN_FORMS:                      # 3 = fused form of the windowed searches, 4 = split form
            T.test_claim_rounds_fuzz_near_thresholds(ctx, orc, wv)
    for form in FORMS:
        T.test_bow_fuzz_near_thresholds(ctx, orc, form)
        T.test_triang_fuzz_crowded_nodes(ctx, orc, form)
    print("seed offset %d ok (%.0f s)" % (1000 * s, time.time() - t0), flush=True)
ctx.close()
print("ALL %d SEEDS OK" % n_seeds)
