#!/usr/bin/env python3
"""Rotation averaging (pgi_rotation_average) against the scipy oracle (sparse direct solves) on RANDOM view graphs: random
neighbourhood graphs (20-700 views, 3-24 neighbours, 1-3 components, noise, up to 30 % outlier edges), sequence graphs (reach 2-6:
the tree path) and dense band graphs (reach 9-24: the two-level path), each also with the two-level solver forced
(PGI_ROTAVG_TWO_LEVEL=2).  Rotations within 1e-5 rad of the oracle's, outer iteration counts within 1, the same bits twice.
Usage (GPU box): soak_rotavg.py [graphs per family, default 15]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("pose-graph-initialization_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import rotavg_oracle as RO
from test_rotavg import sequence_graph
N = int(sys.argv[1]) if len(sys.argv) > 1 else 15
from pyposegraphbuilder import Engine
t_start = time.time()
bad = runs = 0
worst = 0.0


def check(name, src, dst, Rrel, w, V):
    global bad, runs, worst
    Ro, it_o = RO.rotation_average(V, src, dst, Rrel, w)
    for forced in ("1", "2"):
        os.environ["PGI_ROTAVG_TWO_LEVEL"] = forced
        eng = Engine()
        R, it = eng.rotation_average(src, dst, Rrel, w, V)
        R2, it2 = eng.rotation_average(src, dst, Rrel, w, V)
        eng.close()
        d = np.einsum("kij,kmj->kim", R, Ro)
        ang = float(np.arccos(np.clip((np.trace(d, axis1=1, axis2=2) - 1) / 2, -1, 1)).max())
        runs += 1
        worst = max(worst, ang)
        if ang > 1e-5 or abs(int(it) - int(it_o)) > 1 or it2 != it or not np.array_equal(R, R2):
            bad += 1
            print("%s (two-level %s): max diff %.2e rad, %d outer iterations (oracle %d), same bits twice %s" % (
                name, "forced" if forced == "2" else "by rule", ang, it, it_o, it2 == it and np.array_equal(R, R2)))


for seed in range(N):
    rng = np.random.default_rng(3000 + seed)
    V = int(rng.integers(20, 701)); k = int(rng.integers(3, 25)); comps = int(rng.integers(1, 4))
    k = min(k, max(2, V // (2 * comps) - 1))
    src, dst, Rrel, w, Rgt, _ = RO.make_graph(V, k, float(rng.uniform(0, 2.5)), float(rng.uniform(0, 0.3)), seed=seed, components=comps)
    check("random V %d k %d comps %d" % (V, k, comps), src, dst, Rrel, w, V)
    V = int(rng.integers(300, 2500)); reach = int(rng.integers(2, 7)); comps = int(rng.integers(1, 3))
    src, dst, Rrel, w, Rgt = sequence_graph(V, reach, 1.0, 0.05, seed=50 + seed, components=comps)
    check("sequence V %d reach %d comps %d" % (V, reach, comps), src, dst, Rrel, w, V)
    V = int(rng.integers(1100, 3000)); reach = int(rng.integers(9, 25))
    src, dst, Rrel, w, Rgt = sequence_graph(V, reach, 1.0, 0.05, seed=90 + seed)
    check("band V %d reach %d" % (V, reach), src, dst, Rrel, w, V)
    print("... %d graphs per family, %d runs, %d differ, worst %.2e rad (%.0f s)" % (seed + 1, runs, bad, worst, time.time() - t_start), flush=True)
print("rotation-averaging soak: %d runs, %d differ, worst difference from the oracle %.2e rad (%.0f s)" % (runs, bad, worst, time.time() - t_start))
sys.exit(1 if bad else 0)
