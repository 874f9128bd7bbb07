"""Sustained A/B of bf2nn kernel forms: far keyframe pairs (the config-5 triangle is 99.9 % such pairs), 16384 pairs per launch,
each form timed over ~0.4 s of back-to-back launches, rounds interleaved.  ORBM_BF_FORMS=15,16,17 selects the forms."""
import os, sys, time
os.environ.setdefault("ORBM_DIAG_LIB", "1")   # the timing diagnostics live in `make -C orbslam2_amd/csrc diag` only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orbslam2_amd import synth, _lib
from orbslam2_amd.matcher import Context

ctx = Context(0)
L = _lib.load()
kfs = synth.keyframe_pool(seed=5, n_kf=96, n=2000, window=3300)
ids = ctx.upload_frames(kfs)
P = 16384
rng = np.random.default_rng(0)
pa = rng.integers(0, 96, P); pb = (pa + rng.integers(8, 88, P)) % 96      # at least 8 keyframes apart: no shared landmarks
if os.environ.get("ORBM_BF_NEAR") == "1":      # neighbouring keyframes instead: most candidate blocks hold a true match of some query
    pb = (pa + rng.integers(1, 3, P)) % 96
dev = torch.device("cuda:0")
fa = torch.tensor(ids[pa], dtype=torch.int32, device=dev); fb = torch.tensor(ids[pb], dtype=torch.int32, device=dev)
match = torch.empty((P, 2000), dtype=torch.int32, device=dev); counts = torch.empty(P, dtype=torch.int32, device=dev)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
forms = [int(x) for x in os.environ.get("ORBM_BF_FORMS", "13,15,16,18").split(",")]
LAUNCHES = int(os.environ.get("ORBM_BF_LAUNCHES", "80"))
ref = None
for v in forms:
    L.orbm__bf2nn_variant(v)
    ctx.bf2nn_dev(P, fa.data_ptr(), fb.data_ptr(), 0.75, 50, 2000, match.data_ptr(), None, None, counts.data_ptr())
    torch.cuda.synchronize()
    cur = (match.clone(), counts.clone())
    if ref is None: ref = cur
    else: print("form %d identical to form %d:" % (v, forms[0]), bool((cur[0] == ref[0]).all() and (cur[1] == ref[1]).all()), flush=True)
res = {v: [] for v in forms}
for rnd in range(4):
    for v in forms:
        L.orbm__bf2nn_variant(v)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(LAUNCHES):
            ctx.bf2nn_dev(P, fa.data_ptr(), fb.data_ptr(), 0.75, 50, 2000, match.data_ptr(), None, None, counts.data_ptr())
        e1.record(); torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / LAUNCHES)
if 18 in forms:       # the no-ranking diagnostic form leaves the shader clock it saw (MHz) in column 0 of every pair's row
    L.orbm__bf2nn_variant(18)
    for _ in range(20):
        ctx.bf2nn_dev(P, fa.data_ptr(), fb.data_ptr(), 0.75, 50, 2000, match.data_ptr(), None, None, counts.data_ptr())
    torch.cuda.synchronize()
    mhz = match[:, 0].float()
    print("shader clock under the products-only form: mean %.0f MHz, min %.0f, max %.0f" % (mhz.mean().item(), mhz.min().item(), mhz.max().item()), flush=True)
L.orbm__bf2nn_variant(0)
for v in forms:
    t = np.array(res[v])
    print("form %2d  ms/launch %s  -> median %.1f G cmps/s (frac %.3f of 10.07 PF)" % (v, np.round(t, 3), P * 4e6 / np.median(t) / 1e6, P * 4e6 * 512 / np.median(t) / 1e9 / 10066.3), flush=True)
