"""Cost of graph-cut local optimisation on config 2: usage graph_cut_cost.py [libpgi_x.so]"""
import sys, numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pose-graph-initialization_amd"))
from pyposegraphbuilder import _lib as L
if len(sys.argv) > 1:
    L.LIB_PATH = os.path.join(ROOT, "pose-graph-initialization_amd", sys.argv[1])
from pyposegraphbuilder import synthetic as S
from pyposegraphbuilder.engine import Engine
b = S.make_batch(np.arange(10000), 2000)
for lam in (0, 9):
    e = Engine(lo_graph_cut=lam)
    db = e.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, seed=0xB0BA)
    ts = []
    for rep in range(4):
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ed, m = e.estimate_pose_batch(db); z.record(); torch.cuda.synchronize()
        if rep: ts.append(a.elapsed_time(z))
    print("lo_graph_cut", lam, "%.3f ms" % np.median(ts), flush=True)
    e.close()
