#!/usr/bin/env python3
"""How much of K1's time on BASELINE config 2 is the drain at the end of the launch?  Times the same pairs at several
batch sizes (N = 2000) and fits T(P) = tail + P / rate; prints the hypothesis-count distribution too."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-graph-initialization_amd"))
import torch
from pyposegraphbuilder import Engine, synthetic as S
Pmax = int(sys.argv[1]) if len(sys.argv) > 1 else 20480
b = S.make_batch(np.arange(Pmax), 2000)
eng = Engine()
res = []
for P in (1024, 2048, 4096, 8192, 10000, 12288, 16384, 20480):
    if P > Pmax:
        break
    r = int(b["offsets"][P])
    db = eng.upload(b["x1"][:r], b["y1"][:r], b["x2"][:r], b["y2"][:r], b["offsets"][:P + 1], 7.5e-4, seed=0xB0BA)
    eng.estimate_pose_batch(db); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); e, m = eng.estimate_pose_batch(db); z.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(z))
    res.append((P, float(np.median(ts))))
    print("P %6d: %.3f ms  (%.0f edges/s)" % (P, res[-1][1], P / res[-1][1] * 1e3), flush=True)
it = eng.edges_to_numpy(e)["iters"]
print("hypotheses per pair: mean %.1f; share at 64/96/128/160/192/224/256+: %s; max %d" % (
    it.mean(), [round(float(np.mean(it == v)), 3) for v in (64, 96, 128, 160, 192, 224)] + [round(float(np.mean(it >= 256)), 3)], it.max()))
A = np.array([[1.0, p] for p, _ in res if p >= 4096]); y = np.array([t for p, t in res if p >= 4096])
tail, per = np.linalg.lstsq(A, y, rcond=None)[0]
print("fit over P >= 4096: T = %.3f ms + P / (%.0f edges/s): the drain is %.1f %% of the 10 000-pair launch" % (
    tail, 1e3 / per, 100 * tail / (tail + 10000 * per)))
