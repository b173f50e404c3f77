import os, sys
import numpy as np
sys.path.insert(0, "/root/repo/pose-graph-initialization_amd")
import torch
from pyposegraphbuilder import Engine, synthetic as S
P = 8192
eng = Engine()
for rho, N in ((0.5, 2000), (0.7, 2000), (0.3, 2000), (0.5, 600), (0.6, 300), (0.4, 1200)):
    b = S.make_batch(np.arange(P), N, inlier_ratio=rho)
    db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, seed=1)
    line = "rho %.1f N %4d:" % (rho, N)
    for rs in (32, 40, 32, 40):
        eng.set_params(round_size=rs)
        eng.estimate_pose_batch(db); torch.cuda.synchronize()
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3): e, m = eng.estimate_pose_batch(db)
        z.record(); torch.cuda.synchronize()
        got = eng.edges_to_numpy(e)
        errs = [S.rot_err_deg(got["R"][i].reshape(3, 3), b["R"][i]) if got["status"][i] == 1 else np.inf for i in range(P)]
        line += "  rs%d %.3f ms (hyps %.0f, AUC %.4f)" % (rs, a.elapsed_time(z) / 3, got["iters"].mean(), S.auc_at(errs))
    print(line, flush=True)
