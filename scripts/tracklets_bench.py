#!/usr/bin/env python3
"""Device tracklet store (pgi_tracklets_*): time per committed wave and per batched query on a consistent scene --
V views of K keypoints (keypoint permutations of the same physical points), all pairs in random order, waves of W pairs,
60 % of the points matched per pair, 5 % wrong matches, 90 % inliers.  Usage: tracklets_bench.py [V] [K] [W]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-graph-initialization_amd"))
from pyposegraphbuilder.engine import Engine, DeviceTracklets
V = int(sys.argv[1]) if len(sys.argv) > 1 else 24
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
W = int(sys.argv[3]) if len(sys.argv) > 3 else 128
rng = np.random.default_rng(3)
eng = Engine()
dev = eng.device
perm = [rng.permutation(K) for _ in range(V)]
pairs = [(a, b) for a in range(V) for b in range(a + 1, V)]
order = rng.permutation(len(pairs))
trk = DeviceTracklets(eng, V)
total_matches = 0
t_add = t_get = 0.0
for w0 in range(0, len(pairs), W):
    calls, queries = [], []
    for pi in order[w0:w0 + W]:
        a, b = pairs[pi]
        vis = np.nonzero(rng.random(K) < 0.6)[0]
        src, dst = perm[a][vis], perm[b][vis].copy()
        wrong = rng.random(len(vis)) < 0.05
        dst[wrong] = rng.integers(0, K, int(wrong.sum()))
        mask = (rng.random(len(vis)) < 0.9).astype(np.uint8)
        calls.append((a, b, (torch.as_tensor(src.astype(np.int32)).to(dev), torch.as_tensor(dst.astype(np.int32)).to(dev)),
                      torch.as_tensor(mask).to(dev)))
        queries.append((a, b))
        total_matches += int(mask.sum())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = trk.get_correspondences_batch(queries, 5000, raw=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    trk.add_batch(calls)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    t_get += t1 - t0
    t_add += t2 - t1
    info = trk.info()
    print("wave of %d pairs: get %.2f ms (%d correspondences), add %.2f ms, %d launches of the round kernel; store: %d tracks, %d events"
          % (len(calls), 1e3 * (t1 - t0), int(res[2].sum()), 1e3 * (t2 - t1), info["rounds"], info["tracks"], info["events"]))
print("total: %d inlier matches; add %.1f ms = %.1f ns per match; get %.1f ms" % (total_matches, 1e3 * t_add, 1e9 * t_add / total_matches, 1e3 * t_get))
