"""Vocabulary-tree descent at ORBvoc.txt's shape (k = 10, depth 6, 1.1 M nodes): n_frames frames of 2000 features per call
(orbm_frames_compute_bow: descent + FeatureVector build on the device), and one frame alone."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
from orbslam2_amd import synth
from orbslam2_amd.matcher import Context
from oracle import oracle_ffi as orc

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 512
t0 = time.time()
vocab = synth.make_vocabulary_regular()
print("tree: %d nodes, built in %.1f s" % (vocab.n_nodes, time.time() - t0), flush=True)
ctx = Context(0)
vid = ctx.upload_vocabulary(vocab)
rng = np.random.default_rng(3)
frames = []
for k in range(min(n_frames, 64)):
    leaves = rng.integers(vocab.n_nodes - 10 ** 6, vocab.n_nodes, 2000)
    d = synth.observe(rng, vocab.desc[leaves], 0.06)
    frames.append(synth.make_frame(rng, d, np.arange(2000), p_flip=0.0, with_fv=False))
ids = ctx.upload_frames([frames[i % len(frames)] for i in range(n_frames)])
# parity on two frames at this size
word, node = ctx.compute_bow(vid, ids[:2], 4, stride=2000)
for k in range(2):
    w, n, (fn, fo, fi) = orc.transform(vocab, frames[k].desc, 4)
    gn, go, gi = ctx.feature_vector(ids[k], 2000)
    assert np.array_equal(word[k], w) and np.array_equal(node[k], n) and np.array_equal(gn, fn) and np.array_equal(go, fo) and np.array_equal(gi, fi)
print("parity at full size: ok; distinct nodes in frame 0 at levelsup 4: %d, distinct words: %d" % (len(fn), len(set(w.tolist()))), flush=True)
for m in (1, 8, 64, n_frames):
    ts = []
    for rep in range(5):
        ctx.synchronize(); t0 = time.perf_counter()
        ctx.compute_bow(vid, ids[:m], 4, want_ids=False)
        ctx.synchronize(); ts.append(time.perf_counter() - t0)
    ms = min(ts[1:]) * 1e3
    print("%4d frames x 2000 features: %.3f ms  (%.2f M features/s, %.1f G Hamming/s)" % (m, ms, m * 2000 / ms / 1e3, m * 2000 * 60 / ms / 1e6), flush=True)
t0 = time.perf_counter(); orc.transform(vocab, frames[0].desc, 4); t1 = time.perf_counter()
print("CPU restatement, one frame of 2000 features: %.3f ms" % ((t1 - t0) * 1e3))
