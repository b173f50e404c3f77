#!/usr/bin/env python3
"""One-off soak: GPU engine vs CPU oracle, bit for bit, on many ragged random pairs and parameter sets
(usage: soak_parity.py [pairs_per_case]).  Exits non-zero on the first mismatch."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-graph-initialization_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from pyposegraphbuilder import Engine, synthetic as S

P = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
cases = [dict(), dict(round_size=8), dict(lo_iters=0), dict(fixed_budget=96), dict(confidence=0.999, max_iters=300),
         dict(min_inliers=5, vote_all_rows=1), dict(lo_linear_pct=0), dict(guess_mode=1), dict(lo_linear_pct=10, lo_iters=3),
         dict(guess_mode=1, round_size=16, lo_linear_pct=60), dict(sampler=1), dict(sampler=1, max_iters=160, round_size=24),
         dict(lo_graph_cut=9), dict(lo_graph_cut=9, guess_mode=1), dict(lo_graph_cut=40, lo_linear_pct=0, lo_iters=3), dict(lo_graph_cut=3, sampler=1)]
if os.environ.get("SOAK_CASES"):   # e.g. SOAK_CASES=12,13,14,15: only these
    keep = [int(v) for v in os.environ["SOAK_CASES"].split(",")]
    cases = [c if i in keep else None for i, c in enumerate(cases)]
total = 0
t00 = time.time()
for ci, kw in enumerate(cases):
    if kw is None:
        continue
    rng = np.random.default_rng(1000 + ci)
    sizes = rng.choice([5, 6, 7, 9, 16, 33, 64, 65, 100, 127, 128, 129, 200, 257, 400, 700, 1200, 2500], P)
    ids = np.arange(ci * 1000000, ci * 1000000 + P)
    b = S.make_batch(ids, sizes, inlier_ratio=float(rng.choice([0.25, 0.4, 0.6, 0.9])), noise_px=float(rng.choice([0.1, 0.5, 1.5])))
    thr = float(rng.choice([4e-4, 7.5e-4, 2e-3]))
    guesses = has = None
    if ci % 2 == 1:   # chained-pose guesses for a third of the pairs: the true pose, perturbed or garbage
        guesses = np.zeros((P, 12)); has = (rng.random(P) < 0.33).astype(np.uint8)
        for i in np.nonzero(has)[0]:
            R, t = b["R"][i], b["t"][i]
            if rng.random() < 0.3:
                R = S.rodrigues(np.array([0.0, 0, 1]), rng.uniform(0, 0.3)) @ R
            if kw.get("guess_mode") == 1 and rng.random() < 0.7:
                t = rng.standard_normal(3)          # a chained translation is meaningless
            guesses[i, :9], guesses[i, 9:] = R.ravel(), t
    eng = Engine(**kw)
    db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], thr, guesses=guesses, has_guess=has, seed=ci + 7)
    e, m = eng.estimate_pose_batch(db)
    got, masks = eng.edges_to_numpy(e), m.cpu().numpy()
    t0 = time.time()
    exp, emask = O.estimate_pose_batch(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], thr, O.default_params(**kw), ci + 7,
                                       guesses=guesses, has_guess=has, threads=0)
    ok = np.array_equal(masks, emask)
    for k in ("status", "n_inl", "score", "iters", "E", "R", "t", "votes", "cand", "used_guess", "lo_runs"):
        ok &= np.array_equal(got[k], exp[k])
    print("case %d %s: %d pairs, %d rows, thr %.1e: %s  (oracle %.1f s; ok edges %d, guesses used %d)" % (
        ci, kw, P, len(masks), thr, "IDENTICAL" if ok else "MISMATCH", time.time() - t0, int((got["status"] == 1).sum()), int(got["used_guess"].sum())))
    if not ok:
        bad = [i for i in range(P) if any(not np.array_equal(got[k][i], exp[k][i]) for k in ("status", "n_inl", "iters", "E"))]
        print("first mismatching pairs:", bad[:10], "sizes", sizes[bad[:10]])
        sys.exit(1)
    total += P
    eng.close()
# big pairs: every occupancy class of the bucketed launch (LDS rows at 3 and 2 workgroups per CU, rows from L2)
rng = np.random.default_rng(77)
Pb = max(256, P // 20)
sizes = rng.choice([2176, 2177, 3000, 3904, 3905, 6000, 9024, 9025, 12000], Pb)
b = S.make_batch(np.arange(7000000, 7000000 + Pb), sizes, inlier_ratio=0.45, noise_px=0.4)
guesses = np.zeros((Pb, 12)); has = (rng.random(Pb) < 0.25).astype(np.uint8)
for i in np.nonzero(has)[0]:
    guesses[i, :9], guesses[i, 9:] = b["R"][i].ravel(), b["t"][i]
for kwb in (dict(), dict(lo_graph_cut=9)):   # (round 6: also with graph-cut local optimisation -- chains across every rows variant)
    eng = Engine(**kwb)
    db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, guesses=guesses, has_guess=has, seed=99)
    e, m = eng.estimate_pose_batch(db)
    got, masks = eng.edges_to_numpy(e), m.cpu().numpy()
    exp, emask = O.estimate_pose_batch(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, O.default_params(**kwb), 99,
                                       guesses=guesses, has_guess=has, threads=0)
    ok = np.array_equal(masks, emask) and all(np.array_equal(got[k], exp[k]) for k in ("status", "n_inl", "score", "iters", "E", "R", "t", "votes", "cand", "used_guess", "lo_runs"))
    print("big pairs (2176..12000 rows) %s: %d pairs, %d rows: %s" % (kwb, Pb, len(masks), "IDENTICAL" if ok else "MISMATCH"))
    if not ok:
        sys.exit(1)
    total += Pb
    eng.close()
print("estimator soak ok: %d pairs identical in %.0f s" % (total, time.time() - t00))

# ---- matchers: screened descriptor matcher and guided matcher vs the oracle ----
eng = Engine()
rng = np.random.default_rng(4242)
n_match = max(20, P // 100)
sets, pairs = [], []
for q in range(n_match):
    ka, kb = int(rng.integers(2, 700)), int(rng.integers(2, 700))
    A, B, _ = S.make_descriptors(rng, ka, kb, overlap=float(rng.uniform(0, 1)), noise=float(rng.choice([1e-4, 0.02, 0.08])),
                                 duplicates=int(rng.choice([0, 0, 5, 40])))
    if rng.random() < 0.2:
        A, B = (A * 300).round().astype(np.float32), (B * 300).round().astype(np.float32)   # integer-valued, large norms
    sets += [A, B]
    pairs += [(2 * q, 2 * q + 1), (2 * q + 1, 2 * q)]
t0 = time.time()
got = eng.match_descriptors_batch([eng.prepare_descriptors(d) for d in sets], pairs)
for (s_, d_), (gi, gj, gr) in zip(pairs, got):
    oi, oj, orr = O.match_descriptors(sets[s_], sets[d_])
    if not (np.array_equal(gi, oi) and np.array_equal(gj, oj) and np.array_equal(gr, orr)):
        print("MATCH MISMATCH at pair", (s_, d_), len(sets[s_]), len(sets[d_]))
        sys.exit(1)
print("descriptor matcher: %d pairs identical (%.0f s)" % (len(pairs), time.time() - t0))
n_scene = max(4, P // 2000)
n_frag = 0
t0 = time.time()
for q in range(n_scene):
    views, poses, cam = S.make_feature_views(np.random.default_rng(9000 + q), n_views=2, n_points=int(rng.integers(200, 1500)),
                                             n_clutter=int(rng.integers(0, 1500)), desc_noise=float(rng.choice([0.005, 0.012, 0.03])))
    feats = [eng.upload_features(v["xy"], v["desc"], *cam) for v in views]
    R = poses[1][0] @ poses[0][0].T
    t = poses[1][1] - R @ poses[0][1]
    if q % 3 == 2:
        R = S.rodrigues(np.array([0.0, 1, 0]), 0.02) @ R      # a slightly wrong pose: fewer candidates pass the gate
    E = np.zeros(9)
    O.lib().pgo_ref_essential_from_pose(O._p(O.f64(R).ravel()), O._p(O.f64(t)), O._p(E))
    kk = [cam[0], cam[0], cam[1] / 2.0, cam[2] / 2.0]
    F = O.fundamental_from_essential(E, kk, kk)
    for (a_, b_), Rt in (((0, 1), np.r_[R.ravel(), t]),):
        gi, gj, gr = eng.guided_match_batch(feats, [(a_, b_)], Rt[None], max_n=0, n_bins=0)[0]
        oi, oj, orr = O.guided_match(F, views[a_]["xy"], views[b_]["xy"], views[a_]["desc"], views[b_]["desc"])
        if not (np.array_equal(gi, oi) and np.array_equal(gj, oj) and np.array_equal(gr, orr)):
            print("GUIDED MISMATCH at scene", q)
            sys.exit(1)
        # the reference's 45 epipolar bins: device == literal restatement outside the don't-care band at bin edges
        gi, gj, gr = eng.guided_match_batch(feats, [(a_, b_)], Rt[None], max_n=0, n_bins=45)[0]
        size = (int(cam[1]), int(cam[2]))
        oi, oj, orr, frag = O.ref_guided_match_binned(F, views[a_]["xy"], views[b_]["xy"], views[a_]["desc"], views[b_]["desc"], size, size)
        keep_o, keep_g = ~frag[oi].astype(bool), ~frag[gi].astype(bool)
        if not (np.array_equal(gi[keep_g], oi[keep_o]) and np.array_equal(gj[keep_g], oj[keep_o]) and np.array_equal(gr[keep_g], orr[keep_o])):
            print("BINNED GUIDED MISMATCH at scene", q)
            sys.exit(1)
        n_frag += int(frag.sum())
print("guided matcher (exhaustive and 45 bins): %d scenes identical, %d source points in the bin-edge band (%.0f s)" % (n_scene, n_frag, time.time() - t0))
print("soak ok")
