#!/usr/bin/env python3
"""What the round barrier costs in the REAL build (the instrumented one distorts it): config 2's shape with a FIXED budget of 192
hypotheses run as 6 rounds of 32, 3 rounds of 64 and 2 rounds of 96 at each wavefront count -- same hypotheses, same scoring work up
to the bars, fewer workgroup barriers and fewer merge / refit points.  usage: round_barrier_cost.py [P=10000] [N=2000]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pose-graph-initialization_amd"))
import torch
from pyposegraphbuilder import synthetic as S
from pyposegraphbuilder.engine import Engine
P = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
b = S.make_batch(np.arange(P), N)
for nw in (4, 2, 1):
    os.environ["PGI_K1_NW"] = str(nw)
    line = "NW=%d:" % nw
    for rs in (32, 64, 96):
        e = Engine(fixed_budget=192, round_size=rs)
        db = e.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, seed=0xB0BA)
        ts = []
        for rep in range(4):
            a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); ed, m = e.estimate_pose_batch(db); z.record(); torch.cuda.synchronize()
            if rep: ts.append(a.elapsed_time(z))
        got = e.edges_to_numpy(ed)
        line += "  rounds of %d: %.3f ms (refits %.2f)" % (rs, np.median(ts), got["lo_runs"].mean())
        e.close()
    print(line, flush=True)
