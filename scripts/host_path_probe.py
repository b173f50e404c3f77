#!/usr/bin/env python3
"""Host-pointer entry point (pgi_estimate_pose_batch_host) with pageable, registered and hipHostMalloc'd buffers."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-graph-initialization_amd"))
import torch
from pyposegraphbuilder import Engine, synthetic as S, _lib as L
P, N = 10000, 2000
b = S.make_batch(np.arange(P), N)
eng = Engine()
thr = 7.5e-4
def run(x, out, tag):
    eng.estimate_pose_batch_host(*x, b["offsets"], thr, seed=1, out=out)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); eng.estimate_pose_batch_host(*x, b["offsets"], thr, seed=1, out=out); ts.append(time.perf_counter() - t0)
    print("%-28s %.2f ms  %.0f edges/s" % (tag, 1e3 * np.median(ts), P / np.median(ts)), flush=True)
x = [np.ascontiguousarray(b[k], np.float32) for k in ("x1", "y1", "x2", "y2")]
run(x, None, "pageable in, fresh out")
oe, om = np.zeros(P, L.EDGE_DTYPE), np.zeros(P * N, np.uint8)
run(x, (oe, om), "pageable in, reused out")
eng.pin(*x); run(x, (oe, om), "registered in, pageable out")
eng.pin(oe, om); run(x, (oe, om), "registered in+out")
eng.unpin(*x, oe, om)
xp = [torch.from_numpy(a).pin_memory().numpy() for a in x]
oep = torch.zeros(P * 200, dtype=torch.uint8).pin_memory().numpy().view(L.EDGE_DTYPE)
omp = torch.zeros(P * N, dtype=torch.uint8).pin_memory().numpy()
run(xp, (oep, omp), "hipHostMalloc in+out")
run(xp, (oe, om), "hipHostMalloc in, pageable out")
# raw copy rates: one stream vs the four arrays on four streams
nbytes = x[0].nbytes
dev = [torch.empty(x[0].size, dtype=torch.float32, device="cuda") for _ in range(4)]
streams = [torch.cuda.Stream() for _ in range(4)]
for tag, hs in (("pageable", [torch.from_numpy(a) for a in x]), ("pinned", [torch.from_numpy(a) for a in xp])):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(4): dev[k].copy_(hs[k], non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("H2D 4 x %d MB %-9s one stream  : %.2f ms = %.1f GB/s" % (nbytes >> 20, tag, 1e3 * dt, 4 * nbytes / dt / 1e9))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(4):
        with torch.cuda.stream(streams[k]): dev[k].copy_(hs[k], non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("H2D 4 x %d MB %-9s four streams: %.2f ms = %.1f GB/s" % (nbytes >> 20, tag, 1e3 * dt, 4 * nbytes / dt / 1e9))
hm = torch.from_numpy(omp)
dm = torch.empty(P * N, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter(); hm.copy_(dm, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("D2H 20 MB pinned: %.2f ms = %.1f GB/s" % (1e3 * dt, 0.02 / dt))
