#!/usr/bin/env python3
"""BASELINE config 2, ragged variant: 10 000 pairs with N ~ U{50..4000} correspondences (not a bench line)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-graph-initialization_amd"))
import torch
from pyposegraphbuilder import Engine, synthetic as S
P = 10000
ids = np.arange(P)
sizes = S.ragged_sizes(ids)
b = S.make_batch(ids, sizes)
eng = Engine()
db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, seed=1)
eng.estimate_pose_batch(db); torch.cuda.synchronize()
a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(3):
    e, m = eng.estimate_pose_batch(db)
z.record(); torch.cuda.synchronize()
ms = a.elapsed_time(z) / 3
got = eng.edges_to_numpy(e)
ok = got["status"] == 1
errs = [S.rot_err_deg(got["R"][i].reshape(3, 3), b["R"][i]) if ok[i] else np.inf for i in range(P)]
print("ragged N~U{50..4000} (mean %.0f, %.1f M rows): %.3f ms -> %.0f edges/s, %.1f Mrows/s; ok %.4f AUC@5 %.4f" % (
    sizes.mean(), sizes.sum() / 1e6, ms, P / ms * 1e3, sizes.sum() / ms / 1e3, ok.mean(), S.auc_at(errs)))
