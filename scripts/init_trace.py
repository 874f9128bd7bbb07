import sys, os
os.environ["ORBM_DIAG_LIB"] = "1"; os.environ["ORBM_WIN_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.cuda.init()
from orbslam2_amd import synth
from orbslam2_amd.matcher import Context
ctx = Context(0)
f1, f2, prev = synth.init_pair()
ids = ctx.upload_frames([f1, f2])
for i in range(3):
    cnt, m12, p2 = ctx.search_init(ids[0], ids[1], prev, 100, 0.9, True)
print(cnt)
import collections
print("levels of frame 1:", collections.Counter(np.asarray(f1.octave).tolist()))
