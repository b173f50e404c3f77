#!/usr/bin/env python3
"""Soak of the binned guided matcher (pgi_guided_match_batch, n_bins = 45): the pooled scan (list caps 20 / 16 / 4), the tile scan with 2 lanes per source
keypoint and the bin scan (PGI_GUIDED_ANGLE=0) against the literal restatement pgo_ref_guided_match_binned and against each other, bit
for bit, on random scenes: true / perturbed / forward-motion / tiny-baseline / random poses, three image scales, crowded
epipolar lines (sources with > 100 candidates: several rounds).  Usage: soak_guided.py [scenes]; exits non-zero on a mismatch."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-graph-initialization_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from pyposegraphbuilder import Engine, synthetic as S

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
eng = Engine()
rng = np.random.default_rng(777)
t0 = time.time()
n_src = n_match = n_frag = n_crowd = 0
VARIANTS = (("lanes2", {"PGI_GUIDED_LANES": "2"}), ("flat", {}), ("flat_fused", {"PGI_GUIDED_SPLIT": "0"}), ("flat_tiny_arena", {"PGI_GUIDED_ARENA_WORDS": "700"}), ("flat_cap4", {"PGI_GUIDED_CAP": "4"}), ("flat_cap16", {"PGI_GUIDED_CAP": "16"}),
            ("bin_scan", {"PGI_GUIDED_ANGLE": "0"}))
for q in range(n_scenes):
    views, poses, cam = S.make_feature_views(np.random.default_rng(20000 + q), n_views=2, n_points=int(rng.integers(100, 2500)),
                                             n_clutter=int(rng.integers(0, 2500)), desc_noise=float(rng.choice([0.005, 0.012, 0.03])))
    scale = float(rng.choice([1.0, 1.0, 1.0, 8.0, 0.1]))
    cam = (cam[0] * scale, cam[1] * scale, cam[2] * scale)
    views = [dict(v, xy=(v["xy"] * np.float32(scale)).astype(np.float32)) for v in views]
    R = poses[1][0] @ poses[0][0].T
    t = poses[1][1] - R @ poses[0][1]
    kind = q % 6
    if kind == 1:
        R = S.rodrigues(rng.standard_normal(3), float(rng.uniform(0, 0.05))) @ R
    elif kind == 2:
        R, t = np.eye(3), np.array([rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), 1.0])   # forward motion: epipole in the image
    elif kind == 3:
        t = t * 1e-6
    elif kind == 4:
        R, t = S.rodrigues(rng.standard_normal(3), float(rng.uniform(0, 3.0))), rng.standard_normal(3)
    E = np.zeros(9)
    O.lib().pgo_ref_essential_from_pose(O._p(O.f64(R).ravel()), O._p(O.f64(t)), O._p(E))
    kk = [cam[0], cam[0], cam[1] / 2.0, cam[2] / 2.0]
    F = np.asarray(O.fundamental_from_essential(E, kk, kk))
    if kind == 5 and len(views[1]["xy"]) > 400:   # crowd the epipolar lines of two sources
        Fm = F.reshape(3, 3)
        xy2 = views[1]["xy"].astype(np.float64).copy()
        moved = rng.permutation(len(xy2))[:300].reshape(2, 150)
        for src, rows in zip(rng.choice(len(views[0]["xy"]), 2, replace=False), moved):
            l = Fm @ np.r_[views[0]["xy"][src].astype(np.float64), 1.0]
            if abs(l[1]) < 1e-12:
                continue
            x = rng.uniform(0.05 * cam[1], 0.95 * cam[1], 150)
            off = rng.uniform(-0.3, 0.3, 150) / np.hypot(l[0], l[1])
            xy2[rows, 0] = x + off * l[0]
            xy2[rows, 1] = -(l[0] * x + l[2]) / l[1] + off * l[1]
        views[1] = dict(views[1], xy=xy2.astype(np.float32))
        n_crowd += 1
    feats = [eng.upload_features(v["xy"], v["desc"], *cam) for v in views]
    Rt = np.r_[R.ravel(), t][None]
    size = (int(cam[1]), int(cam[2]))
    oi, oj, orr, frag = O.ref_guided_match_binned(F, views[0]["xy"], views[1]["xy"], views[0]["desc"], views[1]["desc"], size, size)
    keep_o = ~frag[oi].astype(bool)
    first = None
    for name, env in VARIANTS:
        for k in ("PGI_GUIDED_LANES", "PGI_GUIDED_ANGLE", "PGI_GUIDED_CAP", "PGI_GUIDED_SPLIT", "PGI_GUIDED_ARENA_WORDS"):
            os.environ.pop(k, None)
        os.environ.update(env)
        for max_n in (0, 100):
            gi, gj, gr = eng.guided_match_batch(feats, [(0, 1)], Rt, max_n=max_n, n_bins=45)[0]
            key = (gi.tobytes(), gj.tobytes(), gr.tobytes())
            if name == "lanes2":
                first = first or {}
                first[max_n] = key
                if max_n == 0:
                    keep_g = ~frag[gi].astype(bool)
                    if not (np.array_equal(gi[keep_g], oi[keep_o]) and np.array_equal(gj[keep_g], oj[keep_o]) and np.array_equal(gr[keep_g], orr[keep_o])):
                        print("MISMATCH against the literal restatement at scene", q, "kind", kind)
                        sys.exit(1)
            elif key != first[max_n]:
                print("MISMATCH between variants at scene", q, "kind", kind, name, "max_n", max_n)
                sys.exit(1)
    n_src += len(views[0]["xy"]); n_match += len(oi); n_frag += int(frag.sum())
print("guided soak: %d scenes (%d with crowded lines), %d source keypoints, %d matches, %d source keypoints in the bin-edge band: pooled scan (two kernels, one kernel, redone wavefronts, caps 16 / 4) == tile scan "
      "(2 lanes) == bin scan == literal restatement (%.0f s)" % (n_scenes, n_crowd, n_src, n_match, n_frag, time.time() - t0))
