#!/usr/bin/env python3
"""Rotation averaging wall time / iterations at BASELINE config 3/4 graph sizes (synthetic view graphs)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-graph-initialization_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import rotavg_oracle as RO
from pyposegraphbuilder import Engine
eng = Engine()
for V, k in ((340, 20), (1500, 20), (5000, 20)):
    src, dst, Rrel, w, Rgt, out = RO.make_graph(V, k, noise_deg=1.0, outlier_frac=0.15, seed=2)
    eng.rotation_average(src, dst, Rrel, w, V)  # warm-up
    t0 = time.perf_counter(); R, iters = eng.rotation_average(src, dst, Rrel, w, V); dt = time.perf_counter() - t0
    err = RO.align_error_deg(R, Rgt)
    line = "V=%5d E=%6d: GPU %.1f ms, %d outer iterations, mean err %.3f deg (max %.3f)" % (V, len(src), 1e3 * dt, iters, err.mean(), err.max())
    if V <= 1500:
        t0 = time.perf_counter(); Ro, it_o = RO.rotation_average(V, src, dst, Rrel, w); dto = time.perf_counter() - t0
        d = np.einsum("kij,kmj->kim", R, Ro); ang = np.arccos(np.clip((np.trace(d, axis1=1, axis2=2) - 1) / 2, -1, 1))
        line += " | scipy oracle %.0f ms, %d iterations, max diff %.1e rad" % (1e3 * dto, it_o, ang.max())
    print(line)

# a band graph (config 4's shape: every view sees its 20 next neighbours along the walk): deep, so the two-level preconditioner runs
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_rotavg import sequence_graph
V = 5000
src, dst, Rrel, w, Rgt = sequence_graph(V, 20, 1.0, 0.05, seed=17)
eng.rotation_average(src, dst, Rrel, w, V)
t0 = time.perf_counter(); R, iters = eng.rotation_average(src, dst, Rrel, w, V); dt = time.perf_counter() - t0
print("band V=%5d E=%6d: GPU %.1f ms, %d outer iterations, mean err %.3f deg" % (V, len(src), 1e3 * dt, iters, RO.align_error_deg(R, Rgt).mean()))
