#!/usr/bin/env python3
"""The wave-scheduled run at correspondence level (PoseGraphBuilder::run through tests/cpp/test_distributed.cpp) against its CPU-only
restatement (oracle/scheduler_oracle.py) on RANDOM scene graphs: 40-140 views, 4-10 neighbours, 0-6 % wrongly retrieved pairs,
waves of 16-128 pairs; reference guesses and rotation-guided re-estimation.  Every scheduler counter equal, the same edges in the
same order with the same scores, poses to 1e-9.  Usage (GPU box): soak_scheduler.py [scenes, default 12]"""
import os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("pose-graph-initialization_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import oracle_lib as O
import scheduler_oracle as SO
from pyposegraphbuilder import synthetic as S
import scene_drivers as SC
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
KEYS = ("pairs_processed", "edges_added", "paths_searched", "paths_found", "touched_nodes", "poses_from_guess", "hypotheses", "waves", "graph_edges",
        "quirk_only_guesses")
t_start = time.time()
bad = runs = 0
for seed in range(N):
    rng = np.random.default_rng(2000 + seed)
    V = int(rng.integers(40, 141)); k = int(rng.integers(4, 11)); wave = int(rng.integers(16, 129))
    g = S.make_scene_graph(V, k=k, seed=100 + seed, outlier_pair_frac=float(rng.uniform(0.0, 0.06)))
    b, sim = g["batch"], SC.pair_similarity(g)
    table, pairs = {}, []
    for e, (i, j) in enumerate(g["pairs"]):
        a, z = int(b["offsets"][e]), int(b["offsets"][e + 1])
        table[(int(i), int(j))] = table[(int(j), int(i))] = float(sim[e])
        pairs.append(dict(src=int(i), dst=int(j), similarity=float(sim[e]), thr=7.5e-4, x1=b["x1"][a:z], y1=b["y1"][a:z], x2=b["x2"][a:z], y2=b["y2"][a:z]))
    lookup = lambda p, q: 1.0 if p == q else table.get((p, q), 0.0)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "scene.bin")
        SC.write_scene(path, g, wave, sim_kind=2)
        for mode in ("waves", "waves_guided"):
            SC.run_ranks([SC.EXE, path, os.path.join(d, mode), mode], 1, extra_env={"PGI_QUIET": "1"})
            stats, edges = SC.read_waves(open(os.path.join(d, mode) + ".0", "rb").read())
            st, ref = SO.run_waves(O, pairs, lookup, V, wave, path_finding=True, rotation_guided=(mode == "waves_guided"))
            runs += 1
            diff = [key for key in KEYS if stats[key] != st[key]]
            if not diff and len(edges) != len(ref):
                diff = ["edge count"]
            if not diff:
                for got, (s_, d_, score, R, t) in zip(edges, ref):
                    if (int(got["src"]), int(got["dst"])) != (s_, d_) or got["score"] != score or np.abs(got["R"].reshape(3, 3) - R).max() > 1e-9 or \
                            np.abs(got["t"] - t).max() > 1e-9:
                        diff.append("edge (%d, %d)" % (s_, d_))
            if diff:
                bad += 1
                print("seed %d (V %d, k %d, wave %d) mode %s DIFFERS: %s" % (seed, V, k, wave, mode, diff[:6]))
    print("... %d scenes, %d runs, %d differ (%.0f s)" % (seed + 1, runs, bad, time.time() - t_start), flush=True)
print("scheduler soak: %d scenes x 2 modes = %d runs, %d differ (%.0f s)" % (N, runs, bad, time.time() - t_start))
sys.exit(1 if bad else 0)
