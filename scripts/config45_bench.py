#!/usr/bin/env python3
"""BASELINE configs 4 and 5 end to end on ONE GPU (world 1), through the C++ host layer (tests/cpp/test_distributed.cpp):
a Trafalgar-sized surrogate (5000 views, ~15 500 candidate pairs, median 100 correspondences) ->
  shard         estimate all pairs + gather + L1/IRLS rotation averaging            (config 4)
  waves         A*-scheduled waves of 4096 pairs, reference-style pose guesses      (config 5, reference guesses)
  waves_guided  the same with rotation-guided re-estimation                         (config 5 as BASELINE names it)
Prints the driver's own wall-clock lines (scene generation and file I/O excluded)."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-graph-initialization_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pyposegraphbuilder import synthetic as S
import test_distributed_gpu as T
V = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
g = S.make_scene_graph(V, k=4, seed=11, outlier_pair_frac=0.03, median_corr=100, min_corr=60, max_corr=400, ring=3)
print("views %d, candidate pairs %d, rows %d" % (V, len(g["pairs"]), int(g["batch"]["offsets"][-1])))
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "scene.bin")
    T.write_scene(path, g, 4096, sim_kind=2)
    for mode in ("shard", "waves", "waves_guided"):
        for rep in range(2):  # second run: warm code objects and allocations
            out = T.run_ranks([T.EXE, path, os.path.join(d, mode), mode], 1)
        print(out[0].strip())
