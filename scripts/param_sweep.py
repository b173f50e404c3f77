#!/usr/bin/env python3
"""Throughput/quality vs estimator parameters (round_size, lo_iters) on BASELINE config 2."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-graph-initialization_amd"))
import torch
from pyposegraphbuilder import Engine, synthetic as S
P, N = 9984, 2000
b = S.make_batch(np.arange(P), N)
eng = Engine()
db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, seed=1)
for rs in (32, 36, 40, 44, 32, 36, 40, 44):
    for lo in (2,):
        eng.set_params(round_size=rs, lo_iters=lo)
        eng.estimate_pose_batch(db); torch.cuda.synchronize()
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            e, m = eng.estimate_pose_batch(db)
        z.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(z) / 3
        got = eng.edges_to_numpy(e)
        errs = [S.rot_err_deg(got["R"][i].reshape(3, 3), b["R"][i]) if got["status"][i] == 1 else np.inf for i in range(P)]
        print("round_size %2d lo %d: %.3f ms %8.0f edges/s  hyps %.1f refits %.2f AUC5 %.4f median %.4f" % (
            rs, lo, ms, P / ms * 1e3, got["iters"].mean(), got["lo_runs"].mean(), S.auc_at(errs), np.median(errs)))
