#!/usr/bin/env python3
"""Throughput / quality vs the number of local-optimisation refits per improvement (lo_iters), several regimes."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-graph-initialization_amd"))
import torch
from pyposegraphbuilder import Engine, synthetic as S
P = 4096
eng = Engine()
for rho, N, noise in ((0.5, 2000, 0.25), (0.3, 2000, 0.25), (0.7, 2000, 0.25), (0.5, 600, 0.25), (0.5, 2000, 0.5), (0.35, 300, 0.25)):
    b = S.make_batch(np.arange(P), N, inlier_ratio=rho, noise_px=noise)
    db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, seed=1)
    for lo in (0, 1, 2, 3):
        eng.set_params(lo_iters=lo)
        eng.estimate_pose_batch(db); torch.cuda.synchronize()
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            e, m = eng.estimate_pose_batch(db)
        z.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(z) / 3
        got = eng.edges_to_numpy(e)
        errs = np.array([S.rot_err_deg(got["R"][i].reshape(3, 3), b["R"][i]) if got["status"][i] == 1 else np.inf for i in range(P)])
        terr = np.array([np.degrees(np.arccos(np.clip(abs(got["t"][i] @ b["t"][i]), -1, 1))) if got["status"][i] == 1 else np.inf for i in range(P)])
        print("rho %.2f N %4d noise %.2f lo %d: %.3f ms %8.0f edges/s hyps %6.1f refits %.2f AUC5 %.4f medR %.4f medT %.3f inl %.1f fail %d" % (
            rho, N, noise, lo, ms, P / ms * 1e3, got["iters"].mean(), got["lo_runs"].mean(), S.auc_at(errs), np.median(errs),
            np.median(terr), got["n_inl"].mean(), int((got["status"] != 1).sum())), flush=True)
