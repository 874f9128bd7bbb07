#!/usr/bin/env python3
"""bench.py -- headline benchmark of the ORB-SLAM2 matcher hot path on MI355X.

Workload: BASELINE.json configs[1] (brute-force 2-NN Hamming + TH_LOW threshold + ratio test, 2000 x 2000 256-bit
descriptors per keyframe pair), batched the way configs[4] asks for it: the pairs are the upper triangle of a
device-resident store of 4500 synthetic keyframes x 2000 features (sliding-window landmark visibility: neighbours overlap
~60 %, far keyframes ~0 %), cut into 32 x 32 keyframe tiles that are dealt block-cyclically to the ranks
(orbm_allpairs_shard).  Every rank holds the whole store (no data-path collective) and walks ITS shard:

  one launch = orbm_allpairs_batch over PAIRS consecutive pairs of the shard: the matching kernel, then the compaction of
               the match rows into 8-byte per-pair summaries + (u16,u16) lists for pairs with >= 20 matches
               (LoopClosing.cc:459); match rows never leave the GPU;
  one step   = LAUNCHES such launches (so that --steps 20 times about half a second, not 30 ms);
  N > 1      = the records of launch k travel to rank 0 over RCCL (torch.distributed "nccl": one fixed-size gather of the
               summaries, then the variable-length lists point to point) on a second stream while launch k + 1 is matched.

Weak scaling: every rank does PAIRS x LAUNCHES pairs per step whatever N is; a rank whose shard is shorter than the run
wraps around to its beginning (the records still travel).  Prints ONE JSON line (contract in the task statement): `value` =
G Hamming comparisons / s over all ranks with inputs resident in HBM; keyframe pairs / s and the secondary paths in `extra`.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_FEAT = 2000
STORE_KF = 4500
TILE = 32
PAIRS_PER_LAUNCH = 16384
LAUNCHES_PER_STEP = 5
LIST_CAP = 1 << 21          # list entries per launch (8 MB); a diagonal-heavy launch of the store holds ~2 x 10^5
ACCEPT = 20                 # LoopClosing.cc:459
RATIO, TH = 0.75, 50
SMALL_KF = 96               # keyframes of the store that the secondary measurements and the CPU baseline use
# The headline kernel computes Hamming distances as exact +-1 dot products on the matrix cores (MX-scaled FP4 MFMA fed from
# the raw descriptor rows): one comparison = a 256-long dot product = 512 FLOP.  Dense FP4 peak: one
# v_mfma_f32_32x32x64_f8f6f4 (2 * 32 * 32 * 64 FLOP) per 32 cycles per SIMD, 256 CUs x 4 SIMDs @ 2.4 GHz
FLOP_PER_CMP = 512.0
PEAK_MFMA_TFLOPS = 256 * 4 * (2 * 32 * 32 * 64 / 32.0) * 2.4e9 / 1e12
# integer-popcount roofline of the vector-ALU kernel (reported beside it): per wave64 comparison one SIMD must issue
# 8 v_xor_b32 (2 cycles each) + 8 v_bcnt_u32_b32 (4 cycles each) = 48 cycles; 256 CUs x 4 SIMDs x 64 lanes @ 2.4 GHz
PEAK_TCMPS = 256 * 4 * 64 * 2.4e9 / 48.0 / 1e12


def build_store(ctx, n_kf, keep_host=SMALL_KF):
    """The config-5 store, generated and uploaded 250 keyframes at a time (same on every rank); the first `keep_host`
    keyframes also stay on the host for the CPU baseline."""
    from orbslam2_amd import synth
    ids, host = [], []
    for s in range(0, n_kf, 250):
        kfs = synth.keyframe_pool(seed=synth.SEED0 + 5 + s, n_kf=min(250, n_kf - s), n=N_FEAT, window=int(N_FEAT / 0.6))
        ids.append(ctx.upload_frames(kfs))
        if len(host) < keep_host:
            host += kfs[: keep_host - len(host)]
    return host, np.concatenate(ids).astype(np.int32)


def pair_list(rng, ids, n_pairs):
    """Pairs of a small store, neighbours (high overlap) and far keyframes mixed (secondary measurements)."""
    a = rng.integers(0, len(ids), n_pairs)
    d = np.where(rng.random(n_pairs) < 0.5, rng.integers(1, 4, n_pairs), rng.integers(4, len(ids), n_pairs))
    b = (a + d) % len(ids)
    return ids[a].astype(np.int32), ids[b].astype(np.int32)


def _all_cores(fn, cores, seconds):
    """fn(k, rng) -> units done by one call, run on `cores` threads for `seconds` (ctypes releases the GIL): units / s"""
    done = [0.0] * cores
    stop = time.perf_counter() + seconds

    def work(k):
        r = np.random.default_rng(100 + k)
        while time.perf_counter() < stop:
            done[k] += fn(k, r)

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(cores)]
    [t.start() for t in th]; [t.join() for t in th]
    return sum(done) / (time.perf_counter() - t0), sum(done)


def cpu_baseline(kfs, n_store):
    """The CPU oracle (restatement of the reference, SWAR popcount as ORBmatcher.cc:2184-2200) on the box's host cores,
    bounded samples of the same workloads.  About 25 s in all."""
    from oracle import oracle_ffi as orc
    from orbslam2_amd import allpairs, synth
    orc.lib()
    flags = orc.timing_build() or "-O2 -msse2 -mpopcnt (prebuilt; no compiler on this machine)"
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    rng = np.random.default_rng(7)
    nk = len(kfs)
    per_pair = float(N_FEAT) * N_FEAT
    # C2, one thread (how the reference runs one matcher call), SWAR then hardware popcount
    t0 = time.perf_counter(); n1 = 0
    while time.perf_counter() - t0 < 2.5:
        i, j = rng.integers(0, nk, 2); orc.bf2nn(kfs[i], kfs[j], RATIO, TH); n1 += 1
    single = n1 * per_pair / (time.perf_counter() - t0)
    orc.set_popcnt(True)
    t0 = time.perf_counter(); n2 = 0
    while time.perf_counter() - t0 < 1.5:
        i, j = rng.integers(0, nk, 2); orc.bf2nn(kfs[i], kfs[j], RATIO, TH); n2 += 1
    single_popcnt = n2 * per_pair / (time.perf_counter() - t0)
    orc.set_popcnt(False)
    # C2 / headline, all cores ("generous CPU": the reference runs every matcher call on one thread)
    def bf(k, r):
        i, j = r.integers(0, nk, 2); orc.bf2nn(kfs[i], kfs[j], RATIO, TH); return 1
    multi_pairs_s, multi_pairs = _all_cores(bf, cores, 8.0)
    # C3: one SearchByProjection call (50k map points, th = 3) on one core, then independent instances on all cores
    f, q = synth.proj_instance(seed=synth.SEED0 + 3, n=N_FEAT, n_queries=50000)
    t1 = time.perf_counter(); _, _, proj_cmps = orc.search_proj_points(f, q, 3.0, 0.8); proj_ms = (time.perf_counter() - t1) * 1e3
    def proj(k, r):
        orc.search_proj_points(f, q, 3.0, 0.8); return 1
    proj_inst_s, _ = _all_cores(proj, cores, 2.0)
    proj_th = {}
    for thv in (1.0, 5.0):                                   # Tracking.cc:1549-1560 calls with th 1, 3 or 5
        t1 = time.perf_counter(); _, _, nc = orc.search_proj_points(f, q, thv, 0.8); proj_th[thv] = ((time.perf_counter() - t1) * 1e3, nc)
    # C1: SearchForInitialization, 2 x 1000 keypoints, window 100, ratio 0.9 (Tracking.cc:768-769), one core: us / call, the
    # reference's Hamming evaluations per call (the oracle's counter) and the match count
    i1, i2, iprev = synth.init_pair()
    c1_cnt, _, _, c1_cmps = orc.search_init(i1, i2, iprev, 100, 0.9, True)
    t1 = time.perf_counter(); c1_n = 0
    while time.perf_counter() - t1 < 1.0:
        orc.search_init(i1, i2, iprev, 100, 0.9, True); c1_n += 1
    c1_us = (time.perf_counter() - t1) * 1e6 / c1_n
    # the two per-frame tracking searches of `extra.track_*`, one core
    fw, qw = synth.window_instance(700, 0, n=N_FEAT, n_queries=1500, th=7.0, stereo=True)
    fp5, qp5 = synth.proj_instance(seed=synth.SEED0 + 9, n=N_FEAT, n_queries=5000)
    trk = {}
    for key, fn in (("track_cur_last_1500q_ms", lambda: orc.search_window(0, fw, qw, 100, True)), ("track_local_map_5000q_ms", lambda: orc.search_proj_points(fp5, qp5, 3.0, 0.8))):
        t1 = time.perf_counter(); nrep = 0
        while time.perf_counter() - t1 < 0.3:
            fn(); nrep += 1
        trk[key] = round((time.perf_counter() - t1) * 1e3 / nrep, 4)
    # C4: SearchByBoW(KF,KF) over the pair list `extra.search_bow_kfkf_*` runs: one core (time + the reference's distance
    # evaluations per pair, the oracle's counter), then all cores, one pair per task
    ia, ib = pair_list(np.random.default_rng(3), np.arange(nk), 8192)
    bow_cmps = 0
    t1 = time.perf_counter()
    for k in range(256):
        bow_cmps += orc.search_bow(1, kfs[ia[k]], kfs[ib[k]], 0.75, True)[2]
    bow_pair_ms = (time.perf_counter() - t1) * 1e3 / 256
    def bow(k, r):
        j = int(r.integers(0, 8192)); orc.search_bow(1, kfs[ia[j]], kfs[ib[j]], 0.75, True); return 1
    bow_pairs_s, _ = _all_cores(bow, cores, 2.0)
    # C5: a 1/1000 random sample of the store's 10 122 750 pairs (regenerated on the host: only the sampled keyframes'
    # chunks), SearchByBoW(KF,KF) on all cores; the brute-force variant on a 1/10000 sample; both EXTRAPOLATED to the triangle
    n_pairs_all = n_store * (n_store - 1) // 2
    srng = np.random.default_rng(5)
    sa = srng.integers(0, n_store, max(1, n_pairs_all // 1000)); sb = srng.integers(0, n_store, len(sa))
    keep = sa != sb; sa, sb = sa[keep], sb[keep]
    chunks = {}
    def kf_of(i):
        c = int(i) // 250
        if c not in chunks:
            chunks[c] = synth.keyframe_pool(seed=synth.SEED0 + 5 + c * 250, n_kf=min(250, n_store - c * 250), n=N_FEAT, window=int(N_FEAT / 0.6))
        return chunks[c][int(i) % 250]
    t1 = time.perf_counter()
    sample = [(kf_of(a), kf_of(b)) for a, b in zip(sa[:4000], sb[:4000])]       # bounded: host generation is the slow part
    gen_s = time.perf_counter() - t1
    pos = [0]; lock = threading.Lock()
    def c5(kind):
        def fn(k, r):
            with lock:
                j = pos[0]; pos[0] += 1
            a, b = sample[j % len(sample)]
            if kind == "bow":
                orc.search_bow(1, a, b, 0.75, True)
            else:
                orc.bf2nn(a, b, RATIO, TH)
            return 1
        return fn
    c5_bow_s, c5_bow_n = _all_cores(c5("bow"), cores, 2.0)
    pos[0] = 0
    c5_bf_s, c5_bf_n = _all_cores(c5("bf"), cores, 3.0)
    return {"value": round(multi_pairs_s * per_pair / 1e9, 4), "unit": "G Hamming cmps/s", "cores": cores, "kind": "port",
            "build": "g++ " + flags, "pairs_per_s": round(multi_pairs_s, 1),
            "single_thread_value": round(single / 1e9, 4), "single_thread_hw_popcnt_value": round(single_popcnt / 1e9, 4),
            "search_by_projection_single_instance_ms": round(proj_ms, 3), "search_by_projection_cmps": proj_cmps,
            "search_by_projection_all_cores_instances_per_s": round(proj_inst_s, 1),
            "search_by_projection_all_cores_queries_per_s": round(proj_inst_s * 50000, 1),
            "search_by_projection_th1_single_instance_ms": round(proj_th[1.0][0], 3), "search_by_projection_th1_cmps": proj_th[1.0][1],
            "search_by_projection_th5_single_instance_ms": round(proj_th[5.0][0], 3), "search_by_projection_th5_cmps": proj_th[5.0][1],
            "c1_search_init_us_per_call": round(c1_us, 2), "c1_search_init_cmps_per_call": c1_cmps, "c1_search_init_matches": c1_cnt,
            "c1_search_init_Gcmps": round(c1_cmps / c1_us / 1e3, 4),
            **trk,
            "c1_note": "BASELINE config 1: SearchForInitialization, 2 frames x 1000 keypoints, 640x480, window 100, ratio 0.9, checkOri, one core",
            "search_bow_kfkf_ms_per_pair": round(bow_pair_ms, 4), "search_bow_kfkf_cmps_per_pair": round(bow_cmps / 256, 1),
            "search_bow_kfkf_all_cores_pairs_per_s": round(bow_pairs_s, 1),
            "c5_all_pairs_search_bow_all_cores_pairs_per_s": round(c5_bow_s, 1),
            "c5_all_pairs_search_bow_extrapolated_s": round(n_pairs_all / c5_bow_s, 1),
            "c5_all_pairs_brute_force_all_cores_pairs_per_s": round(c5_bf_s, 1),
            "c5_all_pairs_brute_force_extrapolated_s": round(n_pairs_all / c5_bf_s, 1),
            "c5_note": f"EXTRAPOLATED from random pairs of the {n_store}-keyframe store ({int(c5_bow_n)} SearchByBoW, {int(c5_bf_n)} brute-force "
                       f"calls over a {len(sample)}-pair sample, {cores} threads) to its {n_pairs_all} pairs; host generation of the sample {gen_s:.1f} s not counted",
            "sample": f"{int(multi_pairs)} random keyframe pairs of the store's first {nk} keyframes ({N_FEAT}x{N_FEAT} brute-force 2-NN, SWAR popcount "
                      f"as ORBmatcher.cc:2184-2200), {cores} threads x 8 s ('generous CPU': the reference runs a matcher call on one thread); "
                      f"single thread {n1} pairs"}


def measured_traffic(n_kf, pairs):
    """HBM bytes per launch of the headline kernel from the committed rocprofv3 counter pass (FETCH_SIZE / WRITE_SIZE in
    their own runs, corrected as MI355X_MICROARCH.md prescribes; scripts/profile_bench.sh + summarize_profile.py): counters
    cannot be read from inside this run, so the line quotes the pass taken on the SAME store size and launch size."""
    import glob
    hit = None
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bf2nn_traffic.json"))):
        try:
            d = json.load(open(fn))
        except Exception:
            continue
        if d.get("store_keyframes") == n_kf and d.get("pairs_per_launch") == pairs:
            hit = (d, os.path.relpath(fn, ROOT))
    return hit


def popcount_form(ctx, torch, dev, ids):
    """The same workload through the vector-ALU xor + popcount kernel (ORBMATCH_BF2NN_KERNEL=valu; identical matches): the form
    BASELINE's north star describes, priced against the integer-popcount roofline."""
    from orbslam2_amd import _lib
    L = _lib.load()
    P = 4096
    fa, fb = pair_list(np.random.default_rng(11), ids, P)
    tfa = torch.tensor(fa, device=dev); tfb = torch.tensor(fb, device=dev)
    m = torch.empty((P, N_FEAT), dtype=torch.int32, device=dev); c = torch.empty(P, dtype=torch.int32, device=dev)
    ms = []
    try:
        L.orbm__bf2nn_variant(11)
        for rep in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ctx.bf2nn_dev(P, tfa.data_ptr(), tfb.data_ptr(), RATIO, TH, N_FEAT, m.data_ptr(), None, None, c.data_ptr())
            e1.record(); torch.cuda.synchronize()
            if rep:
                ms.append(e0.elapsed_time(e1))
    finally:
        L.orbm__bf2nn_variant(0)
    kms = float(np.mean(ms))
    tcmps = P * float(N_FEAT) * N_FEAT / (kms * 1e-3) / 1e12
    return {"bound": "valu-int-popcount", "kernel": "bf2nn_kernel<2,4,0>", "kernel_ms": round(kms, 4), "achieved": round(tcmps, 4),
            "peak": round(PEAK_TCMPS, 4), "unit": "T Hamming cmps/s", "frac": round(tcmps / PEAK_TCMPS, 4),
            "pairs_per_s": round(P / (kms * 1e-3), 1),
            "frac_of_survey_peak": round(tcmps / 2.4576, 4),
            "note": "4096 mixed pairs of the store's first 96 keyframes; peak = 256 CUs x 4 SIMDs x 64 lanes x 2.4 GHz / (8 xor x 2 cycles + 8 bcnt x 4 "
                    "cycles) per comparison (issue rates measured with scripts/ubench_valu*.hip: v_xor is double-rate); SURVEY 8(d) prices all "
                    "16 lane-ops at 64 lanes/clk/CU = 2.4576 T cmps/s (frac_of_survey_peak)"}


def secondary(ctx, torch, dev, ids):
    """Short measurements of the other kernels (parity-tested configs 3/4), reported in `extra`."""
    from orbslam2_amd import synth
    out = {}
    ev = lambda: torch.cuda.Event(enable_timing=True)
    # config 4: SearchByBoW(KF,KF) over pairs of the store
    rng = np.random.default_rng(3)
    fa, fb = pair_list(rng, ids, 8192)
    tfa = torch.tensor(fa, device=dev); tfb = torch.tensor(fb, device=dev)
    match = torch.empty((8192, N_FEAT), dtype=torch.int32, device=dev); counts = torch.empty(8192, dtype=torch.int32, device=dev)
    best = 1e30
    for rep in range(4):
        e0, e1 = ev(), ev(); e0.record()
        ctx.search_bow_dev(1, 8192, tfa.data_ptr(), tfb.data_ptr(), 0.75, True, N_FEAT, match.data_ptr(), counts.data_ptr())
        e1.record(); torch.cuda.synchronize()
        if rep:
            best = min(best, e0.elapsed_time(e1))
    out["search_bow_kfkf_pairs_per_s"] = round(8192 / (best * 1e-3), 1)
    # config 2 again with three more forms of the kernel (identical matches, tests/test_gpu_bf2nn.py): xor + popcount with early
    # abandon, the int8 MFMA form and round 1's FP4 MFMA form (pre-expanded images staged through LDS)
    from orbslam2_amd import _lib
    L = _lib.load()
    P2 = 4096
    fa2, fb2 = pair_list(np.random.default_rng(11), ids, P2)
    tfa2 = torch.tensor(fa2, device=dev); tfb2 = torch.tensor(fb2, device=dev)
    m2 = torch.empty((P2, N_FEAT), dtype=torch.int32, device=dev); c2 = torch.empty(P2, dtype=torch.int32, device=dev)
    for variant, name in ((8, "valu_early_abandon"), (9, "mfma_int8"), (10, "mfma_fp4_lds_staged")):
        try:
            L.orbm__bf2nn_variant(variant)
            best = 1e30
            for rep in range(4):
                e0, e1 = ev(), ev(); e0.record()
                ctx.bf2nn_dev(P2, tfa2.data_ptr(), tfb2.data_ptr(), RATIO, TH, N_FEAT, m2.data_ptr(), None, None, c2.data_ptr())
                e1.record(); torch.cuda.synchronize()
                if rep:
                    best = min(best, e0.elapsed_time(e1))
        finally:
            L.orbm__bf2nn_variant(0)
        out[f"bf2nn_{name}_pairs_per_s"] = round(P2 / (best * 1e-3), 1)
        out[f"bf2nn_{name}_Gcmps"] = round(P2 * N_FEAT * N_FEAT / (best * 1e-3) / 1e9, 1)
    # config 4 as stated: ONE frame against 64 candidate keyframes (relocalisation / loop closure), SearchByBoW(KF, F)
    cand = ids[rng.choice(len(ids), 64, replace=False)].astype(np.int32)
    tca = torch.tensor(cand, device=dev); tcb = torch.tensor(np.full(64, ids[0], np.int32), device=dev)
    best = 1e30
    for rep in range(5):
        e0, e1 = ev(), ev(); e0.record()
        ctx.search_bow_dev(0, 64, tca.data_ptr(), tcb.data_ptr(), 0.75, True, N_FEAT, match.data_ptr(), counts.data_ptr())
        e1.record(); torch.cuda.synchronize()
        if rep:
            best = min(best, e0.elapsed_time(e1))
    out["search_bow_64_candidates_ms"] = round(best, 4)
    out["search_bow_64_candidates_pairs_per_s"] = round(64 / (best * 1e-3), 1)
    # config 3: SearchByProjection (th = 3), 50k map points into a 2000-keypoint frame.  Batch of instances with
    # distinct query sets (permutations of one set, so the queries really stream from HBM), then one instance alone.
    from orbslam2_amd.views import ProjQueries
    n_inst, n_sets, NQ = 256, 256, 50000
    f, q = synth.proj_instance(seed=synth.SEED0 + 3, n=N_FEAT, n_queries=NQ)
    prng = np.random.default_rng(0)
    sets = []
    for k in range(n_sets):                                  # 256 distinct sets = 0.68 GB of queries: more than the 256 MB Infinity Cache
        p = prng.permutation(q.n)
        sets.append(ProjQueries(desc=q.desc[p], proj_x=q.proj_x[p], proj_y=q.proj_y[p], proj_xr=q.proj_xr[p], level=q.level[p],
                                view_cos=q.view_cos[p], flags=q.flags[p]))
    fid = ctx.upload_frames([f] * 8); qid = ctx.upload_proj_queries(sets)
    del sets
    tfr = torch.tensor(np.resize(fid, n_inst), device=dev); tqs = torch.tensor(np.resize(qid, n_inst), device=dev)
    owner = torch.empty((n_inst, N_FEAT), dtype=torch.int32, device=dev); cnt = torch.empty(n_inst, dtype=torch.int32, device=dev)
    # algorithmic bytes per instance: per query 32 B descriptor + 5 words + flag, the frame (49 B / keypoint + grid), owner out
    bytes_inst = NQ * (32 + 20 + 1) + N_FEAT * 49 + 3073 * 4 + N_FEAT * 4
    for thv, ni in ((3.0, n_inst), (3.0, 1), (1.0, n_inst), (5.0, n_inst)):
        best = 1e30
        for rep in range(4):
            e0, e1 = ev(), ev(); e0.record()
            ctx.search_proj_points_dev(ni, tfr.data_ptr(), tqs.data_ptr(), thv, 0.8, None, N_FEAT, owner.data_ptr(), cnt.data_ptr())
            e1.record(); torch.cuda.synchronize()
            if rep:
                best = min(best, e0.elapsed_time(e1))
        tag = "" if thv == 3.0 else f"_th{int(thv)}"
        if ni > 1:
            out[f"search_by_projection{tag}_queries_per_s"] = round(ni * NQ / (best * 1e-3), 1)
            out[f"search_by_projection{tag}_algorithmic_GBps"] = round(ni * bytes_inst / (best * 1e-3) / 1e9, 1)
            out[f"search_by_projection{tag}_hbm_frac"] = round(ni * bytes_inst / (best * 1e-3) / 8e12, 4)
            out[f"search_by_projection{tag}_batch_ms"] = round(best, 4)
            if thv == 3.0:
                out["search_by_projection_note"] = (f"{ni} instances x {NQ} queries -> 2000-kp frame per launch, th=3 (_th1 / _th5: th = 1 / 5), {n_sets} "
                                                    f"distinct query sets; {bytes_inst} algorithmic bytes per instance; 8 TB/s HBM peak")
        else:
            out["search_by_projection_single_instance_ms"] = round(best, 4)
    ctx.free_frames(fid); ctx.free_queries(qid)
    # config 1: SearchForInitialization, 2 x 1000 keypoints, window 100 (host lists in and out: a single call as Tracking.cc:768 makes it)
    i1, i2, iprev = synth.init_pair()
    iid = ctx.upload_frames([i1, i2])
    best = 1e30
    for rep in range(6):
        ctx.synchronize(); t0 = time.perf_counter()
        c1 = ctx.search_init(iid[0], iid[1], iprev, 100, 0.9, True)[0]
        dt1 = time.perf_counter() - t0
        if rep:
            best = min(best, dt1)
    out["c1_search_init_ms"] = round(best * 1e3, 4)
    out["c1_search_init_matches"] = int(c1)
    ctx.free_frames(iid)
    # the two searches a tracking thread issues per frame, as single calls with host lists and outputs (resident frame and query
    # set: the numpy -> ctypes marshalling of a fresh set is this wrapper's, not the library's; the C++ class sends the set with
    # the call): SearchByProjection(Cur, Last) 1500 points th 7, SearchByProjection(F, 5000 map points) th 3
    fw, qw = synth.window_instance(700, 0, n=N_FEAT, n_queries=1500, th=7.0, stereo=True)
    fp, qp = synth.proj_instance(seed=synth.SEED0 + 9, n=N_FEAT, n_queries=5000)
    tid2 = ctx.upload_frames([fw, fp]); wq = ctx.upload_win_queries([qw]); pq = ctx.upload_proj_queries([qp])
    for key, fn in (("track_cur_last_1500q_ms", lambda: ctx.search_window(0, tid2[:1], wq, 100, True, N_FEAT)),
                    ("track_local_map_5000q_ms", lambda: ctx.search_proj_points(tid2[1:], pq, 3.0, 0.8, N_FEAT))):
        best = 1e30
        for rep in range(8):
            ctx.synchronize(); t0 = time.perf_counter()
            fn()
            dt1 = time.perf_counter() - t0
            if rep:
                best = min(best, dt1)
        out[key] = round(best * 1e3, 4)
    ctx.free_frames(tid2); ctx.free_queries(wq); ctx.free_queries(pq)
    # ComputeBoW on the device: descent of a k = 10, depth-6 tree (ORBvoc.txt's shape, 1.1 M nodes, synthetic) with every
    # descriptor of 96 keyframes + FeatureVector build, in one call; then one frame alone.  (Runs last: it replaces the
    # FeatureVectors of these keyframes with those of the synthetic tree.)
    vid = ctx.upload_vocabulary(synth.make_vocabulary_regular())
    for nfr in (len(ids), 1):
        best = 1e30
        for rep in range(4):
            ctx.synchronize(); t0 = time.perf_counter()
            ctx.compute_bow(vid, ids[:nfr], 4, want_ids=False)
            ctx.synchronize(); best = min(best, time.perf_counter() - t0) if rep else best
        if nfr > 1:
            out["compute_bow_features_per_s"] = round(nfr * N_FEAT / best, 1)
        else:
            out["compute_bow_single_frame_ms"] = round(best * 1e3, 4)
    ctx.free_vocabulary(vid)
    # KeyFrameDatabase scoring: one query BowVector against 4096 keyframe BowVectors (~1900 words each), DBoW2 L1 score
    vecs = synth.bow_vectors(n_vec=64)
    bid = ctx.upload_bowvecs(vecs)
    tva = torch.tensor(np.full(4096, bid[0], np.int32), device=dev); tvb = torch.tensor(np.resize(bid, 4096).astype(np.int32), device=dev)
    sc = torch.empty(4096, dtype=torch.float64, device=dev); cm = torch.empty(4096, dtype=torch.int32, device=dev)
    best = 1e30
    for rep in range(4):
        e0, e1 = ev(), ev(); e0.record()
        ctx.bow_score_dev(4096, tva.data_ptr(), tvb.data_ptr(), sc.data_ptr(), cm.data_ptr())
        e1.record(); torch.cuda.synchronize()
        if rep:
            best = min(best, e0.elapsed_time(e1))
    out["bow_l1_score_pairs_per_s"] = round(4096 / (best * 1e-3), 1)
    out["bow_l1_score_4096_keyframes_ms"] = round(best, 4)
    ctx.free_bowvecs(bid)
    return out


def full_triangle(ctx, torch, dev, fa_all, fb_all, pid_all, summ, lists, kind):
    """Config 5 as stated, on this one GPU: every pair of the store once (extra; the N-GPU job is this loop over each rank's shard)."""
    n = fa_all.numel()
    P = PAIRS_PER_LAUNCH
    acc = torch.zeros((), dtype=torch.int64, device=dev); tot = torch.zeros((), dtype=torch.int64, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(0, n, P):
        e = min(n, s + P)
        ctx.allpairs_batch_dev(kind, e - s, fa_all.data_ptr() + 4 * s, fb_all.data_ptr() + 4 * s, pid_all.data_ptr() + 4 * s, 0.75, TH, True,
                               N_FEAT, ACCEPT, summ.data_ptr(), lists.data_ptr(), LIST_CAP, summ.data_ptr() + 8 * P)
        hdr = summ[P]
        acc += hdr & 0xFFFFFFFF; tot += hdr >> 32
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt, int(acc), int(tot)


def self_launch(argv, n_gpus):
    """`python3 bench.py --gpus N` without a launcher: start N fresh rank processes through torch.distributed.run (one per
    GPU, rendezvous on 127.0.0.1) and relay rank 0's JSON line.  This process never touches the GPU (no torch import, no
    orbm_create): the ranks are children, never an exec of a process that holds a device."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:                      # the ranks' stderr goes straight through; stdout is relayed, the JSON line kept
        if out.startswith("{") and '"metric"' in out:
            line = out.strip()
        else:
            sys.stdout.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc != 0 or line is None:
        raise SystemExit(rc if rc != 0 else 1)
    raise SystemExit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=PAIRS_PER_LAUNCH, help="pairs per launch")
    ap.add_argument("--launches", type=int, default=LAUNCHES_PER_STEP, help="launches per step")
    ap.add_argument("--keyframes", type=int, default=STORE_KF)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--launch-probe", action="store_true",
                    help="ranks only rendezvous (gloo, no GPU) and rank 0 prints how many it saw: the CPU test of the self-launch branch")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(sys.argv[1:], args.gpus)      # does not return

    import torch
    import torch.distributed as dist
    from orbslam2_amd import allpairs
    from orbslam2_amd.matcher import Context

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.launch_probe:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo")
            t = torch.tensor([rank], dtype=torch.int64)
            seen = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(seen, t)
            ranks = sorted(int(x.item()) for x in seen)
            dist.barrier(); dist.destroy_process_group()
        else:
            ranks = [0]
        if rank == 0:
            print(json.dumps({"metric": "launch-probe", "n_gpus": world, "ranks_seen": len(ranks), "ranks": ranks}), flush=True)
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    # rehearsal only: ORBM_BENCH_BACKEND=gloo lets N ranks share one GPU (records hop through host memory)
    backend = os.environ.get("ORBM_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    comm_dev = dev if backend == "nccl" else torch.device("cpu")

    ctx = Context(dev_index)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    n_kf = args.keyframes
    t_setup = time.perf_counter()
    kfs, ids = build_store(ctx, n_kf)                       # same store on every rank (replicated)
    ctx.bf2nn_prepare()                                     # a no-op for the default kernel (it reads the raw rows); builds the images of the other MFMA forms
    P, L = args.pairs, args.launches
    pa, pb = allpairs.shard_pairs(n_kf, rank, world, TILE)  # this rank's block-cyclic share of the triangle
    n_mine = len(pa)
    if n_mine < P:
        raise SystemExit(f"rank {rank}: shard of {n_mine} pairs is shorter than one launch of {P}")
    tid = torch.tensor(ids, device=dev)
    fa_all = tid[torch.tensor(pa, device=dev).long()].contiguous()
    fb_all = tid[torch.tensor(pb, device=dev).long()].contiguous()
    pid_all = torch.tensor(allpairs.pair_ids(pa, pb, n_kf).view(np.int32), device=dev)
    n_full = n_mine // P                                    # whole launches in the shard; launch j takes window j % n_full
    setup_s = time.perf_counter() - t_setup
    summ = [torch.empty(P + 1, dtype=torch.int64, device=dev) for _ in range(2)]
    lists = [torch.empty(LIST_CAP, dtype=torch.int32, device=dev) for _ in range(2)]
    hdr_pin = [torch.empty(1, dtype=torch.int64).pin_memory() for _ in range(2)]
    done_ev = [torch.cuda.Event() for _ in range(2)]
    comm_ev = [None, None]
    comm_stream = torch.cuda.Stream(device=dev)
    compute_stream = torch.cuda.current_stream()
    n_launch_total = (args.steps + args.warmup) * L
    # rank 0 keeps what arrives: every summary of the run, lists in a ring
    sink_summ = torch.empty(n_launch_total * world * P, dtype=torch.int64, device=comm_dev) if rank == 0 else None
    SINK_LISTS = 1 << 26
    sink_lists = torch.empty(SINK_LISTS, dtype=torch.int32, device=comm_dev) if rank == 0 else None
    sink = {"summ": 0, "lists": 0, "entries": 0, "bytes_wire": 0, "overflow": 0}

    def take(r, s_t, l_t):
        n = s_t.numel()
        sink_summ[sink["summ"]: sink["summ"] + n].copy_(s_t, non_blocking=True); sink["summ"] += n
        m = l_t.numel()
        if m:
            if sink["lists"] + m > SINK_LISTS:
                sink["lists"] = 0
            sink_lists[sink["lists"]: sink["lists"] + m].copy_(l_t, non_blocking=True); sink["lists"] += m
        sink["entries"] += m
        if r:
            sink["bytes_wire"] += 8 * (n + 1) + 4 * m

    ex = allpairs.RecordExchange(torch, dist, rank, world, P, comm_dev, take if rank == 0 else None, LIST_CAP)
    kern_ev = []
    pending = []

    def launch(j, timed):
        b = j & 1
        if comm_ev[b] is not None:
            compute_stream.wait_event(comm_ev[b])           # the records of launch j - 2 have left these buffers
        s = (j % n_full) * P
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.allpairs_batch_dev(0, P, fa_all.data_ptr() + 4 * s, fb_all.data_ptr() + 4 * s, pid_all.data_ptr() + 4 * s, RATIO, TH, True,
                               N_FEAT, ACCEPT, summ[b].data_ptr(), lists[b].data_ptr(), LIST_CAP, summ[b].data_ptr() + 8 * P)
        e1.record()
        hdr_pin[b].copy_(summ[b][P:], non_blocking=True)
        done_ev[b].record()
        if timed:
            kern_ev.append((e0, e1))

    def ship(j):
        """records of launch j -> rank 0, on the communication stream (launch j + 1 is already queued on the compute stream)"""
        b = j & 1
        done_ev[b].synchronize()
        entries = int(hdr_pin[b].item()) >> 32
        if entries > LIST_CAP:
            sink["overflow"] += 1
            entries = LIST_CAP
        with torch.cuda.stream(comm_stream):
            s_t, l_t = summ[b], lists[b]
            if comm_dev.type == "cpu":
                s_t, l_t = s_t.cpu(), l_t[:entries].cpu()
            ex.submit(s_t, l_t, entries)
            comm_ev[b] = torch.cuda.Event(); comm_ev[b].record()

    def run(first, n):
        for j in range(first, first + n):
            launch(j, first >= args.warmup * L)
            if j > first:
                ship(j - 1)
        ship(first + n - 1)
        comm_stream.synchronize()

    # The headline is the form that multiplies EVERY bit of EVERY comparison (bf2nn variant 22, sign-free operands).  The
    # library's default for a batch like this one defers the last 64 bits of a comparison until a 192-bit lower bound reaches
    # the acceptance bound (variant 0: identical matches, about three of four products executed on unrelated pairs); that
    # form is timed beside it on the same launches (roofline_deferred_form) and is never `value`.
    from orbslam2_amd import _lib as _lib_hd
    _lib_hd.load().orbm__bf2nn_variant(22)
    run(0, args.warmup * L)
    torch.cuda.synchronize()
    # the matching kernel alone, by HIP events the library records around it on its own stream (orbm_kernel_timing): what the
    # roofline figure is about, and what the rocprofv3 kernel trace of this command measures as well
    import ctypes
    kt_on = ctx._L.orbm_kernel_timing(ctx._h, args.steps * L) == 0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.warmup * L, args.steps * L)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # what every rank did, as rank 0 sees it through the process group: proof that N ranks took part
    mine = torch.tensor([rank, P * L * args.steps, n_mine, dev_index], dtype=torch.int64, device=comm_dev)
    if world > 1:
        seen = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(seen, mine)
        per_rank = [[int(v) for v in t.cpu().tolist()] for t in seen]
        ranks_seen = dist.get_world_size()
    else:
        per_rank = [[int(v) for v in mine.cpu().tolist()]]
        ranks_seen = 1
    _lib_hd.load().orbm__bf2nn_variant(0)
    call_ms = float(np.mean([a.elapsed_time(b) for a, b in kern_ev])) if kern_ev else float("nan")
    kms = call_ms
    if kt_on:
        buf = (ctypes.c_float * (args.steps * L))()
        got_n = ctx._L.orbm_kernel_times(ctx._h, buf, args.steps * L)
        ctx._L.orbm_kernel_timing(ctx._h, 0)
        if got_n > 0:
            kms = float(np.mean(buf[: min(got_n, args.steps * L)]))
    cmps_per_launch = float(P) * N_FEAT * N_FEAT
    pairs_total = float(sum(p_[1] for p_ in per_rank))
    if rank == 0:
        achieved_tcmps = cmps_per_launch / (kms * 1e-3) / 1e12
        achieved = achieved_tcmps * FLOP_PER_CMP
        got = sink_summ[: sink["summ"]]
        n_valid = int((got & 0xFFFFFFFF != 0xFFFFFFFF).sum())
        matches_seen = int(((got >> 32) & 0xFFFF).sum())
        res = {
            "metric": "G Hamming cmps/s + keyframe-pairs matched/s (2k feats/frame), 1/2/4/8 GPUs",
            "value": round(pairs_total * N_FEAT * N_FEAT / dt / 1e9, 2),
            "unit": "G Hamming cmps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp4 e2m1 (candidate bits complemented as 0 / 1, query bits as 0 / 2: exact) x fp4 e2m1 -> f32 accumulate (exact integers)", "data": "synthetic",
            "config": {"workload": "brute-force 2-NN Hamming + TH_LOW + ratio test, 2000x2000 256-bit descriptors per keyframe pair "
                                   "(BASELINE configs[1]), batched over the pair triangle of the config-5 store (BASELINE configs[4])",
                       "store_keyframes": n_kf, "features_per_keyframe": N_FEAT, "pairs_in_triangle": n_kf * (n_kf - 1) // 2,
                       "pairs_per_launch": P, "launches_per_step": L, "pairs_per_step_per_gpu": P * L, "ratio": RATIO, "th": TH,
                       "accept_threshold": ACCEPT, "tile": TILE,
                       "parallelism": (f"pair triangle dealt block-cyclically ({TILE}x{TILE} keyframe tiles) to {world} ranks, store replicated, "
                                       f"compact records to rank 0 over RCCL (gather of summaries + point-to-point lists) on a second stream")
                                      if world > 1 else "1 GPU (rank 0's shard = the whole triangle)"},
            "pairs_per_s": round(pairs_total / dt, 1),
            "timed_region_s": round(dt, 4),
            "backend": ("nccl (RCCL)" if backend == "nccl" else backend) if world > 1 else "none (1 rank)",
            "records": {"ranks_seen": ranks_seen,
                        "per_rank": [{"rank": r, "pairs_timed": p_, "shard_pairs": s_, "device": d_} for r, p_, s_, d_ in per_rank],
                        "pairs_received": n_valid, "matches_in_summaries": matches_seen, "list_entries_received": sink["entries"],
                        "bytes_over_the_wire": sink["bytes_wire"],
                        "wire_bytes_per_remote_pair": round(sink["bytes_wire"] / max(1, n_valid * (world - 1) / world), 2) if world > 1 else 0.0,
                        "record_bytes_per_pair": round((8 * n_valid + 4 * sink["entries"]) / max(1, n_valid), 2),
                        "list_buffer_overflows": sink["overflow"],
                        "shard_pairs_rank0": n_mine, "shard_wraps": round(n_launch_total / n_full, 2)},
            "setup_s": round(setup_s, 1),
            "roofline": {"bound": "mfma", "achieved": round(achieved, 1), "peak": round(PEAK_MFMA_TFLOPS, 1),
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_MFMA_TFLOPS, 4), "traffic": None,
                         "kernel": "bf2nn_mfma4r_kernel<false,4,2>", "kernel_ms": round(kms, 4), "call_ms": round(call_ms, 4),
                         "flop_per_cmp": FLOP_PER_CMP, "achieved_T_cmps_per_s": round(achieved_tcmps, 4),
                         "note": "every bit of every comparison is multiplied (bf2nn variant 22; the library's default for this batch defers the last 64 bits: roofline_deferred_form). Hamming distance + popcount(query) = popcount(candidate) + dot of the sign-free images (candidate bits "
                                 "complemented as 0 / 1, query bits as 0 / 2; the +-1 images of rounds 1-3, distance = (256 - dot) / 2, cost "
                                 "the chip 10 % of its clock: extra.bf2nn_operand_ab); algorithmic FLOP per launch = pairs x 2000 x 2000 "
                                 "x 512; peak = 256 CUs x 4 SIMDs x 4096 FLOP/cycle (v_mfma_f32_32x32x64_f8f6f4, FP4) x 2.4 GHz. "
                                 "kernel_ms = HIP events the library records around the matching kernel on its stream (orbm_kernel_timing), "
                                 "averaged over every launch of the timed region; call_ms = events around the whole "
                                 "orbm_allpairs_batch call (+ its counter memset and the two record-compaction kernels). traffic: "
                                 "PMC counters cannot be read inside this run; the field quotes the committed rocprofv3 counter pass "
                                 "taken on the same store size and pairs per launch (traffic_source), null when there is none. "
                                 "vs_integer_popcount_roofline = achieved cmps/s over the 3.277 T cmps/s that bound the xor + popcount "
                                 "form of this kernel (roofline_popcount_form: that form, same run)"},
        }
        res["roofline"]["vs_integer_popcount_roofline"] = round(achieved_tcmps / PEAK_TCMPS, 3)
        tr = measured_traffic(n_kf, P)
        if tr:
            res["roofline"]["traffic"] = round(tr[0]["hbm_bytes_per_launch"], 1)
            res["roofline"]["traffic_unit"] = "HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, KiB -> bytes)"
            res["roofline"]["traffic_source"] = tr[1]
            res["roofline"]["traffic_vs_naive"] = round(tr[0]["hbm_bytes_per_launch"] / (P * (2.0 * N_FEAT * 32 + N_FEAT * 4 + 4)), 4)
        small = ids[:SMALL_KF]
        try:
            res["roofline_popcount_form"] = popcount_form(ctx, torch, dev, small)
        except Exception as e:
            res["roofline_popcount_form"] = {"error": repr(e)}
        if not args.no_extra:
            try:
                ex_out = {}
                if world == 1:
                    for kind, name in ((1, "search_bow"), (0, "brute_force")):
                        s_dt, s_acc, s_ent = full_triangle(ctx, torch, dev, fa_all, fb_all, pid_all, summ[0], lists[0], kind)
                        ex_out[f"c5_all_pairs_{name}_s"] = round(s_dt, 3)
                        ex_out[f"c5_all_pairs_{name}_pairs_per_s"] = round(n_mine / s_dt, 1)
                        ex_out[f"c5_all_pairs_{name}_accepted_pairs"] = s_acc
                        ex_out[f"c5_all_pairs_{name}_record_bytes_per_pair"] = round((8 * n_mine + 4 * s_ent) / n_mine, 2)
                    # the same launches with +-1 operands (what rounds 1-3 multiplied) and with the sign-free operands, sustained
                    # and interleaved: same matches, different clock under the power limit
                    from orbslam2_amd import _lib as _lib_ab
                    ab = {17: [], 22: [], 0: []}
                    try:
                        for rep in range(2):
                            for v in (17, 22, 0):
                                _lib_ab.load().orbm__bf2nn_variant(v)
                                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                                e0.record()
                                for j in range(40):
                                    s0 = (j % n_full) * P
                                    ctx.allpairs_batch_dev(0, P, fa_all.data_ptr() + 4 * s0, fb_all.data_ptr() + 4 * s0, pid_all.data_ptr() + 4 * s0, RATIO,
                                                           TH, True, N_FEAT, ACCEPT, summ[0].data_ptr(), lists[0].data_ptr(), LIST_CAP, summ[0].data_ptr() + 8 * P)
                                e1.record(); torch.cuda.synchronize()
                                ab[v].append(e0.elapsed_time(e1) / 40)
                    finally:
                        _lib_ab.load().orbm__bf2nn_variant(0)
                    ex_out["bf2nn_operand_ab"] = {"pm1_operands_Gcmps": round(cmps_per_launch / (min(ab[17]) * 1e-3) / 1e9, 1),
                                                  "sign_free_operands_Gcmps": round(cmps_per_launch / (min(ab[22]) * 1e-3) / 1e9, 1),
                                                  "sign_free_last_64_bits_deferred_Gcmps": round(cmps_per_launch / (min(ab[0]) * 1e-3) / 1e9, 1),
                                                  "note": "40 launches of 16384 pairs per form and repetition, two repetitions interleaved, whole calls"}
                    d_ms = min(ab[0])
                    res["roofline_deferred_form"] = {
                        "kernel": "bf2nn_mfma4r_kernel<false,4,4>", "call_ms": round(d_ms, 4),
                        "comparisons_decided_T_per_s": round(cmps_per_launch / (d_ms * 1e-3) / 1e12, 4),
                        "pairs_per_s": round(P / (d_ms * 1e-3), 1),
                        "vs_headline_form": round(min(ab[22]) / d_ms, 3),
                        "note": "the library's default for large match-only batches whose acceptance bound is <= 72 (th 50, ratio 0.75: 67): three "
                                "of the four products of a 32-row block give 192-bit partial distances = lower bounds; the fourth group is "
                                "issued only for blocks where some partial distance is below its lane's bound (about one block in twenty on "
                                "unrelated pairs). Identical matches (tests: variants 22 / 24 against the oracle). It executes fewer products "
                                "than the headline form, so it is reported here in comparisons DECIDED per second and is not priced "
                                "against the MFMA peak; extra.c5_all_pairs_brute_force_* run through it (the default)"}
                ex_out.update(secondary(ctx, torch, dev, small))
                res["extra"] = ex_out
            except Exception as e:  # secondary numbers must never break the headline line
                res["extra"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(kfs, n_kf)
            ex_d, cb = res.get("extra", {}), res["cpu_baseline"]
            if "search_bow_kfkf_pairs_per_s" in ex_d:   # BoW-grouped search in the metric's other unit: the reference's evaluations/s
                ex_d["search_bow_kfkf_Gcmps"] = round(ex_d["search_bow_kfkf_pairs_per_s"] * cb["search_bow_kfkf_cmps_per_pair"] / 1e9, 2)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
