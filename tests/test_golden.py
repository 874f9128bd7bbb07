"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle):
 - CPU: the oracle still reproduces them (regression of the checker);
 - GPU: the HIP path reproduces them through the C ABI, bit for bit."""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)
from orbslam2_amd import synth  # noqa: E402
from orbslam2_amd.views import ProjQueries, WinQueries  # noqa: E402


def load(name):
    return np.load(os.path.join(HERE, "golden", name))


def unpack_q(z, prefix, cls, keys):
    return cls(**{k: z[f"{prefix}_{k}"] for k in keys if f"{prefix}_{k}" in z})


class OracleSide:
    def __init__(self, orc):
        self.orc = orc

    def bf(self, a, b): c, m, bd, sd, _ = self.orc.bf2nn(a, b, 0.75, 50); return c, m, bd, sd
    def bow(self, v, a, b, ori): c, m, _ = self.orc.search_bow(v, a, b, 0.75, ori); return c, m
    def tri(self, a, b, f12, ep, os_, ori): c, m, _ = self.orc.search_triang(a, b, f12, ep, os_, ori); return c, m
    def proj(self, f, q, th): c, o, _ = self.orc.search_proj_points(f, q, th, 0.8); return c, o
    def win(self, mode, f, q, dth, ori): c, o, _ = self.orc.search_window(mode, f, q, dth, ori); return c, o
    def init(self, a, b, prev): c, m, p, _ = self.orc.search_init(a, b, prev, 100, 0.9, True); return c, m, p
    def free(self, f, q, dth, chi2): return self.orc.search_window_free(f, q, dth, chi2)[0]
    def distinct(self, off, desc): return np.array([self.orc.distinctive_descriptor(desc[off[i]:off[i + 1]]) for i in range(len(off) - 1)])
    def stereo(self, l, r, md): i, d, _ = self.orc.stereo_match(l, r, 480, md); return i, d
    def transform(self, vocab, feat, up): w, n, fv = self.orc.transform(vocab, feat, up); return (w, n) + tuple(fv)
    def bow_scores(self, vecs):
        r = [[self.orc.bow_score(a, b) for b in vecs] for a in vecs]
        return np.array([[x[0] for x in row] for row in r]), np.array([[x[1] for x in row] for row in r], np.int32)


class GpuSide:
    def __init__(self, ctx):
        self.ctx = ctx

    def _up(self, *frames):
        return self.ctx.upload_frames(list(frames))

    def bf(self, a, b):
        ids = self._up(a, b); c, m, bd, sd = self.ctx.bf2nn([ids[0]], [ids[1]], 0.75, 50, stride=a.n); self.ctx.free_frames(ids)
        return c[0], m[0], bd[0], sd[0]

    def bow(self, v, a, b, ori):
        ids = self._up(a, b); n = b.n if v == 0 else a.n
        c, m = self.ctx.search_bow(v, [ids[0]], [ids[1]], 0.75, ori, stride=n); self.ctx.free_frames(ids)
        return c[0], m[0]

    def tri(self, a, b, f12, ep, os_, ori):
        ids = self._up(a, b); c, m = self.ctx.search_triang([ids[0]], [ids[1]], f12[None], ep[None], os_, ori, stride=a.n)
        self.ctx.free_frames(ids); return c[0], m[0]

    def proj(self, f, q, th):
        ids = self._up(f); qid = self.ctx.upload_proj_queries([q])
        c, o = self.ctx.search_proj_points(ids, qid, th, 0.8, stride=f.n); self.ctx.free_frames(ids); self.ctx.free_queries(qid)
        return c[0], o[0]

    def win(self, mode, f, q, dth, ori):
        ids = self._up(f); qid = self.ctx.upload_win_queries([q])
        c, o = self.ctx.search_window(mode, ids, qid, dth, ori, stride=f.n); self.ctx.free_frames(ids); self.ctx.free_queries(qid)
        return c[0], o[0]

    def init(self, a, b, prev):
        ids = self._up(a, b); r = self.ctx.search_init(ids[0], ids[1], prev, 100, 0.9, True); self.ctx.free_frames(ids)
        return r

    def free(self, f, q, dth, chi2):
        ids = self._up(f); qid = self.ctx.upload_win_queries([q])
        b = self.ctx.search_window_free(ids, qid, dth, chi2, stride_q=q.n); self.ctx.free_frames(ids); self.ctx.free_queries(qid)
        return b[0]

    def distinct(self, off, desc): return self.ctx.distinctive_descriptors(off, desc)

    def stereo(self, l, r, md):
        ids = self._up(l, r); i, d = self.ctx.stereo_match([ids[0]], [ids[1]], [md], stride=l.n); self.ctx.free_frames(ids)
        return i[0], d[0]


    def transform(self, vocab, feat, up):
        rng = np.random.default_rng(0)
        ids = self._up(synth.make_frame(rng, feat, np.arange(len(feat)), p_flip=0.0, with_fv=False))
        vid = self.ctx.upload_vocabulary(vocab)
        w, n = self.ctx.compute_bow(vid, ids, up, stride=len(feat))
        fv = self.ctx.feature_vector(ids[0], len(feat))
        self.ctx.free_vocabulary(vid); self.ctx.free_frames(ids)
        return (w[0], n[0]) + tuple(fv)


    def bow_scores(self, vecs):
        ids = self.ctx.upload_bowvecs(vecs)
        a, b = np.meshgrid(np.arange(len(vecs)), np.arange(len(vecs)), indexing="ij")
        s, c = self.ctx.bow_score(ids[a.ravel()], ids[b.ravel()])
        self.ctx.free_bowvecs(ids)
        return s.reshape(a.shape), c.reshape(a.shape)


def check_all(side):
    z = load("pair_bf_bow.npz")
    a, b = mg.unpack_frame(z, "a"), mg.unpack_frame(z, "b")
    c, m, bd, sd = side.bf(a, b)
    assert c == z["bf_count"] and np.array_equal(m, z["bf_match"]) and np.array_equal(bd, z["bf_best"]) and np.array_equal(sd, z["bf_second"])
    for v in (0, 1):
        for ori in (0, 1):
            c, m = side.bow(v, a, b, bool(ori))
            assert c == z[f"bow{v}{ori}_count"] and np.array_equal(m, z[f"bow{v}{ori}_match"])
    z = load("triang.npz")
    a, b = mg.unpack_frame(z, "a"), mg.unpack_frame(z, "b")
    for os_ in (0, 1):
        for ori in (0, 1):
            c, m = side.tri(a, b, z["f12"], z["epipole"], bool(os_), bool(ori))
            assert c == z[f"tri{os_}{ori}_count"] and np.array_equal(m, z[f"tri{os_}{ori}_match"])
    z = load("proj_points.npz")
    f, q = mg.unpack_frame(z, "f"), unpack_q(z, "q", ProjQueries, mg.PQ_KEYS)
    for th in (1.0, 3.0):
        c, o = side.proj(f, q, th)
        assert c == z[f"proj{int(th)}_count"] and np.array_equal(o, z[f"proj{int(th)}_owner"])
    z = load("window_modes.npz")
    for mode, dth in ((0, 100), (1, 64), (2, 50)):
        f, q = mg.unpack_frame(z, f"f{mode}"), unpack_q(z, f"q{mode}", WinQueries, mg.WQ_KEYS)
        for ori in (0, 1):
            c, o = side.win(mode, f, q, dth, bool(ori))
            assert c == z[f"win{mode}{ori}_count"] and np.array_equal(o, z[f"win{mode}{ori}_owner"])
    z = load("init.npz")
    a, b = mg.unpack_frame(z, "a"), mg.unpack_frame(z, "b")
    c, m, p = side.init(a, b, z["prev"])
    assert c == z["init_count"] and np.array_equal(m, z["init_match"]) and np.array_equal(p, z["init_prev"])


    z = load("next_rows.npz")
    f, q = mg.unpack_frame(z, "wf"), unpack_q(z, "wq", WinQueries, mg.WQ_KEYS)
    for chi2, dth in ((0, 100), (1, 50)):
        assert np.array_equal(side.free(f, q, dth, bool(chi2)), z[f"free{chi2}_best"])
    assert (z["free0_best"] >= 0).sum() > 100 and (z["free1_best"] >= 0).sum() > 20
    assert np.array_equal(side.distinct(z["dd_off"], z["dd_desc"]), z["dd_best"])
    l, r = mg.unpack_frame(z, "sl"), mg.unpack_frame(z, "sr")
    i, d = side.stereo(l, r, float(z["stereo_max_d"]))
    assert np.array_equal(i, z["stereo_idx"]) and np.array_equal(d, z["stereo_dist"]) and (z["stereo_dist"] < 75).sum() > 100

    z = load("vocab_transform.npz")
    vocab = mg.unpack_vocab(z)
    for up in (1, 4):
        got = side.transform(vocab, z["feat"], up)
        for g, key in zip(got, ("word", "node", "fv_node", "fv_off", "fv_idx")):
            assert np.array_equal(g, z[f"t{up}_{key}"]), (up, key)
    z2 = load("bow_score.npz")
    from orbslam2_amd.views import BowVecData
    vecs = [BowVecData(z2[f"bv{i}_word"], z2[f"bv{i}_value"]) for i in range(int(z2["n_vec"]))]
    sc, cm = side.bow_scores(vecs)
    assert np.array_equal(sc.view(np.uint64), z2["score"].view(np.uint64)) and np.array_equal(cm, z2["common"])
    assert (z2["common"] > 20).sum() > 30 and (z2["score"] > 0.05).sum() > 30
    assert (z["t1_node"] < 0).sum() > 20 and len(z["t1_fv_node"]) > 10 and z["t4_fv_node"].tolist() == [0]


def test_oracle_reproduces_golden(orc):
    check_all(OracleSide(orc))


@pytest.mark.gpu
def test_hip_path_reproduces_golden(ctx):
    check_all(GpuSide(ctx))
