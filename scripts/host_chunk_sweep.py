import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo/pose-graph-initialization_amd")
import torch
from pyposegraphbuilder import Engine, synthetic as S, _lib as L
P, N = 10000, 2000
b = S.make_batch(np.arange(P), N)
eng = Engine()
thr = 7.5e-4
SEED = int(os.environ.get("SEED", "1"), 0)
x = [torch.from_numpy(np.ascontiguousarray(b[k], np.float32)).pin_memory().numpy() for k in ("x1", "y1", "x2", "y2")]
oe = torch.zeros(P * 200, dtype=torch.uint8).pin_memory().numpy().view(L.EDGE_DTYPE)
om = torch.zeros(P * N, dtype=torch.uint8).pin_memory().numpy()
for cfg in sys.argv[1:]:
    os.environ["PGI_HOST_CHUNKS"] = cfg
    for _ in range(3): eng.estimate_pose_batch_host(*x, b["offsets"], thr, seed=SEED, out=(oe, om))
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); eng.estimate_pose_batch_host(*x, b["offsets"], thr, seed=SEED, out=(oe, om)); ts.append(time.perf_counter() - t0)
    print("chunks %-6s %.2f ms  %.0f edges/s" % (cfg, 1e3 * np.median(ts), P / np.median(ts)), flush=True)
