"""Phase times of fused_kernel for one tracking-sized call (DIAG library, ORBM_WIN_TRACE): a5 and the LastFrame overload."""
import sys, os
os.environ["ORBM_DIAG_LIB"] = "1"; os.environ["ORBM_WIN_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.cuda.init()
from orbslam2_amd import synth, _lib
from orbslam2_amd.matcher import Context
ctx = Context(0)
_lib.load().orbm__window_variant(3)
f, q = synth.window_instance(700, 0, n=2000, n_queries=1500, th=7.0, stereo=True)
fid = ctx.upload_frames([f]); qid = ctx.upload_win_queries([q])
for i in range(3):
    counts, owner = ctx.search_window(0, fid, qid, 100, True, 2000)
print("LastFrame 1500 q:", counts)
fp, qp = synth.proj_instance(seed=synth.SEED0 + 9, n=2000, n_queries=5000)
fpid = ctx.upload_frames([fp]); qpid = ctx.upload_proj_queries([qp])
for i in range(3):
    counts, owner = ctx.search_proj_points(fpid, qpid, 3.0, 0.8, 2000)
print("proj 5000 q:", counts)

# ---- the split form (collect_kernel + window_kernel): stamps of instance 0
import ctypes
L = _lib.load()
L.orbm__window_variant(4)
buf = (ctypes.c_ulonglong * 128)()
def split_stamps(name, fn):
    for i in range(3):
        L.orbm__window_stamps(buf, 1)
        fn()
    L.orbm__window_stamps(buf, 0)
    t = np.array(buf[:], dtype=np.float64) / 100.0
    c = t[0:6] - t[0]
    print("%s collect (us from its start): counted %.1f | scanned %.1f | scattered+staged %.1f | classified %.1f | walked %.1f" % (name, c[1], c[2], c[3], c[4], c[5]))
    w0 = t[16]
    rounds = [t[21 + r] - w0 for r in range(40) if t[21 + r] > 0]
    print("%s claims (us from its start): state set up %.1f | live queries listed %.1f (%d live) | rounds begin %.1f | rounds end at %s | tail %.1f | end %.1f; gap collect end -> claims start %.1f"
          % (name, t[17] - w0, t[18] - w0, int(buf[70]), t[20] - w0, " ".join("%.1f" % r for r in rounds), t[62] - w0, t[63] - w0, w0 - t[5]))
split_stamps("LastFrame 1500 q", lambda: ctx.search_window(0, fid, qid, 100, True, 2000))
split_stamps("proj 5000 q", lambda: ctx.search_proj_points(fpid, qpid, 3.0, 0.8, 2000))
