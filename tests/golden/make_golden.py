"""Generates tests/golden/*.npz: small seeded inputs plus the CPU ORACLE's outputs for them.

The reference tree has no runnable matcher, tests or fixtures, so these are regression vectors
of the oracle (parity unpinned), not reference outputs.  Re-run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from orbslam2_amd import synth  # noqa: E402
from orbslam2_amd.views import FrameData, ProjQueries, WinQueries  # noqa: E402
from oracle import oracle_ffi as orc  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
FRAME_KEYS = ["desc", "x", "y", "angle", "octave", "uright", "mp_flags", "fv_node", "fv_off", "fv_idx"]


def pack_frame(prefix, f):
    d = {}
    for k in FRAME_KEYS:
        v = getattr(f, k)
        if v is not None:
            d[f"{prefix}_{k}"] = v
    return d


def unpack_frame(z, prefix):
    kw = {k: z[f"{prefix}_{k}"] for k in FRAME_KEYS if f"{prefix}_{k}" in z}
    return FrameData(**kw)


def pack_q(prefix, q, keys):
    return {f"{prefix}_{k}": getattr(q, k) for k in keys if getattr(q, k) is not None}


PQ_KEYS = ["desc", "proj_x", "proj_y", "proj_xr", "level", "view_cos", "flags"]
WQ_KEYS = ["desc", "u", "v", "radius", "min_level", "max_level", "ur", "angle", "flags"]


def make_next_rows():
    """a12 (claim-free best-1, with and without the chi-square gate) and the SURVEY 8(f) rank-4 rows."""
    d = {}
    f, q = synth.window_instance(31, 2, n=300, n_queries=700, stereo=True, th=8.0)
    rng = np.random.default_rng(31)
    q.ur = np.where(rng.random(q.n) < 0.8, q.u - rng.uniform(0, 40, q.n), -1.0).astype(np.float32)
    d.update(pack_frame("wf", f)); d.update(pack_q("wq", q, WQ_KEYS))
    for chi2, dth in ((0, 100), (1, 50)):
        best, _ = orc.search_window_free(f, q, dth, bool(chi2))
        d[f"free{chi2}_best"] = best
    sizes = [0, 1, 2, 3, 7, 20, 64, 65, 100]
    off, rows = [0], []
    for n in sizes:
        lm = synth.random_descriptors(rng, 1)
        rows.append(np.concatenate([synth.observe(rng, lm, 0.08) for _ in range(n)]) if n else np.zeros((0, 32), np.uint8))
        off.append(off[-1] + n)
    d["dd_off"] = np.array(off, np.int32); d["dd_desc"] = np.concatenate(rows)
    d["dd_best"] = np.array([orc.distinctive_descriptor(r) for r in rows], np.int32)
    n = 400
    lm = synth.random_descriptors(rng, n)
    xl = rng.uniform(16, 624, n).astype(np.float32); yl = rng.uniform(16, 464, n).astype(np.float32)
    left = synth.make_frame(rng, lm, np.arange(n), x=xl, y=yl, with_fv=False)
    perm = rng.permutation(n)
    right = synth.make_frame(rng, lm, perm, x=(xl - rng.uniform(-2, 50, n).astype(np.float32))[perm],
                             y=(yl + rng.normal(0, 1.0, n).astype(np.float32))[perm],
                             octave=np.clip(left.octave.astype(np.int32) + rng.integers(-2, 3, n), 0, 7).astype(left.octave.dtype)[perm],
                             with_fv=False)
    d.update(pack_frame("sl", left)); d.update(pack_frame("sr", right))
    si, sd, _ = orc.stereo_match(left, right, 480, 45.0)
    d.update(stereo_idx=si, stereo_dist=sd, stereo_max_d=np.float32(45.0))
    np.savez_compressed(os.path.join(OUT, "next_rows.npz"), **d)
    make_vocab()


VOCAB_KEYS = ["child_off", "child_idx", "desc", "word_id", "weight"]


def unpack_vocab(z):
    from orbslam2_amd.views import VocabData
    return VocabData(depth=int(z["voc_depth"]), **{k: z[f"voc_{k}"] for k in VOCAB_KEYS})


def make_vocab():
    """An irregular tree with sibling ties and stopped words, 600 features, the restated DBoW2 descent at two levelsup."""
    vocab = synth.make_vocabulary(seed=41, k=5, depth=4, irregular=True, tie_frac=0.2, stop_frac=0.15)
    rng = np.random.default_rng(42)
    feat = synth.vocabulary_features(rng, vocab, 600)
    d = {f"voc_{k}": getattr(vocab, k) for k in VOCAB_KEYS}
    d.update(voc_depth=np.int32(vocab.depth), feat=feat)
    for up in (1, 4):
        w, n, (fn, fo, fi) = orc.transform(vocab, feat, up)
        d.update({f"t{up}_word": w, f"t{up}_node": n, f"t{up}_fv_node": fn, f"t{up}_fv_off": fo, f"t{up}_fv_idx": fi})
    np.savez_compressed(os.path.join(OUT, "vocab_transform.npz"), **d)
    # KeyFrameDatabase similarity: 12 BowVectors, every ordered pair, DBoW2 L1 score (double) + common words
    vecs = synth.bow_vectors(seed=43, n_vec=12, n_words_vocab=5000, n_feat=300, group=4)
    d = {"n_vec": np.int32(len(vecs))}
    for i, v in enumerate(vecs):
        d[f"bv{i}_word"] = v.word; d[f"bv{i}_value"] = v.value
    sc = np.zeros((12, 12)); cm = np.zeros((12, 12), np.int32)
    for i in range(12):
        for j in range(12):
            sc[i, j], cm[i, j] = orc.bow_score(vecs[i], vecs[j])
    d.update(score=sc, common=cm)
    np.savez_compressed(os.path.join(OUT, "bow_score.npz"), **d)


def main():
    if "--next-only" in sys.argv:
        make_next_rows()
        return
    kfs = synth.keyframe_pool(seed=11, n_kf=2, n=300, window=450, mp_valid_frac=0.9, stereo_frac=0.3, bad_frac=0.05)
    a, b = kfs
    d = {}
    d.update(pack_frame("a", a)); d.update(pack_frame("b", b))
    c, m, bd, sd, _ = orc.bf2nn(a, b, 0.75, 50)
    d.update(bf_count=c, bf_match=m, bf_best=bd, bf_second=sd)
    for variant in (0, 1):
        for ori in (0, 1):
            c, m, _ = orc.search_bow(variant, a, b, 0.75, bool(ori))
            d[f"bow{variant}{ori}_count"] = c; d[f"bow{variant}{ori}_match"] = m
    np.savez_compressed(os.path.join(OUT, "pair_bf_bow.npz"), **d)

    f1, f2, f12, ep = synth.triang_pair(12, n=400)
    d = {}
    d.update(pack_frame("a", f1)); d.update(pack_frame("b", f2)); d.update(f12=f12, epipole=ep)
    for os_ in (0, 1):
        for ori in (0, 1):
            c, m, _ = orc.search_triang(f1, f2, f12, ep, bool(os_), bool(ori))
            d[f"tri{os_}{ori}_count"] = c; d[f"tri{os_}{ori}_match"] = m
    np.savez_compressed(os.path.join(OUT, "triang.npz"), **d)

    f, q = synth.proj_instance(seed=13, n=300, n_queries=1500, stereo=True, claimed_frac=0.15, obs_frac=0.7, valid_frac=0.9)
    d = {}
    d.update(pack_frame("f", f)); d.update(pack_q("q", q, PQ_KEYS))
    for th in (1.0, 3.0):
        c, o, _ = orc.search_proj_points(f, q, th, 0.8)
        d[f"proj{int(th)}_count"] = c; d[f"proj{int(th)}_owner"] = o
    np.savez_compressed(os.path.join(OUT, "proj_points.npz"), **d)

    d = {}
    for mode, dist_th in ((0, 100), (1, 64), (2, 50)):
        f, q = synth.window_instance(14 + mode, mode, n=300, n_queries=800, stereo=(mode == 0))
        d.update(pack_frame(f"f{mode}", f)); d.update(pack_q(f"q{mode}", q, WQ_KEYS))
        for ori in (0, 1):
            c, o, _ = orc.search_window(mode, f, q, dist_th, bool(ori))
            d[f"win{mode}{ori}_count"] = c; d[f"win{mode}{ori}_owner"] = o
    np.savez_compressed(os.path.join(OUT, "window_modes.npz"), **d)

    f1, f2, prev = synth.init_pair(seed=17, n=500)
    d = {}
    d.update(pack_frame("a", f1)); d.update(pack_frame("b", f2)); d["prev"] = prev
    c, m, p, _ = orc.search_init(f1, f2, prev, 100, 0.9, True)
    d.update(init_count=c, init_match=m, init_prev=p)
    np.savez_compressed(os.path.join(OUT, "init.npz"), **d)
    make_next_rows()
    for fn in sorted(os.listdir(OUT)):
        if fn.endswith(".npz"):
            print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
