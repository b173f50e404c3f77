#!/usr/bin/env python3
"""Phase breakdown of estimate_pose_kernel by timing variants (fixed budget, N sweep, LO on/off)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-graph-initialization_amd"))
import torch
from pyposegraphbuilder import Engine, synthetic as S

P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
eng = Engine()
def timeit(db, reps=3):
    eng.estimate_pose_batch(db); torch.cuda.synchronize()
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        e, m = eng.estimate_pose_batch(db)
    z.record(); torch.cuda.synchronize()
    return a.elapsed_time(z) / reps, eng.edges_to_numpy(e)
for N in (64, 512, 1024, 2048, 4096):
    b = S.make_batch(np.arange(P), N)
    db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, seed=1)
    for fb, lo in ((256, 0), (256, 2), (0, 2)):
        eng.set_params(fixed_budget=fb, lo_iters=lo)
        ms, got = timeit(db)
        print("N=%5d budget=%4d lo=%d : %8.3f ms  %9.0f edges/s  hyps=%.1f lo_runs=%.2f us/pair=%.2f" % (
            N, fb, lo, ms, P / ms * 1e3, got["iters"].mean(), got["lo_runs"].mean(), ms * 1e3 / P))
