import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo/pose-graph-initialization_amd")
import torch
from pyposegraphbuilder import Engine, synthetic as S, _lib as L
P, N = 10000, 2000
b = S.make_batch(np.arange(P), N)
eng = Engine()
thr = 7.5e-4
x = [torch.from_numpy(np.ascontiguousarray(b[k], np.float32)).pin_memory().numpy() for k in ("x1", "y1", "x2", "y2")]
oe = torch.zeros(P * 200, dtype=torch.uint8).pin_memory().numpy().view(L.EDGE_DTYPE)
om = torch.zeros(P * N, dtype=torch.uint8).pin_memory().numpy()
def measure(tag):
    for _ in range(3): eng.estimate_pose_batch_host(*x, b["offsets"], thr, seed=1, out=(oe, om))
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); eng.estimate_pose_batch_host(*x, b["offsets"], thr, seed=1, out=(oe, om)); ts.append(time.perf_counter() - t0)
    print("%-40s %.2f ms" % (tag, 1e3 * np.median(ts)), flush=True)
measure("fresh engine")
db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], thr, seed=1)
for _ in range(10): e, m = eng.estimate_pose_batch(db)
torch.cuda.synchronize()
measure("after resident runs (torch stream bound)")
for _ in range(3): eng.estimate_pose_batch_host(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], thr, seed=1)
measure("after pageable host runs")
thr_arr = np.full(P, thr)
ts=[]
for _ in range(5):
    t0 = time.perf_counter(); eng.estimate_pose_batch_host(*x, b["offsets"], thr_arr, seed=1, pair_id_base=0, out=(oe, om)); ts.append(time.perf_counter() - t0)
print("thr array %.2f ms" % (1e3*np.median(ts)))
# the bench's own sequence: fixed-budget run, sequential upload run, then the host path again
eng.set_params(fixed_budget=256)
edges = torch.empty((P, 200), dtype=torch.uint8, device=eng.device)
masks = torch.empty(P * N, dtype=torch.uint8, device=eng.device)
eng.estimate_pose_batch(db, edges, masks); torch.cuda.synchronize()
eng.set_params(fixed_budget=0)
db2 = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], thr, seed=1)
e2, m2 = eng.estimate_pose_batch(db2, edges, masks)
_ = e2.cpu(), m2.cpu()
torch.cuda.synchronize()
del db2
measure("after the bench's sequential leg")
px = [torch.from_numpy(np.ascontiguousarray(b[k], np.float32)).pin_memory().numpy() for k in ("x1", "y1", "x2", "y2")]
pe = torch.zeros(P * 200, dtype=torch.uint8).pin_memory().numpy().view(L.EDGE_DTYPE)
pm = torch.zeros(P * N, dtype=torch.uint8).pin_memory().numpy()
x, oe, om = px, pe, pm
measure("page-locked buffers allocated late")
