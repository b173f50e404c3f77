#!/usr/bin/env python3
"""Times BASELINE config 2 with alternative builds of libpgi (compiler-flag experiments): lib paths on the command line."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-graph-initialization_amd"))
import subprocess
if len(sys.argv) > 2:  # one child per library (a process can load only one libpgi)
    for lib in sys.argv[1:]:
        subprocess.call([sys.executable, os.path.abspath(__file__), lib])
    raise SystemExit(0)
from pyposegraphbuilder import _lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from pyposegraphbuilder import Engine, synthetic as S
P, N = 10000, 2000
b = S.make_batch(np.arange(P), N)
eng = Engine()
db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, seed=0xB0BA)
for _ in range(2):
    eng.estimate_pose_batch(db)
torch.cuda.synchronize()
a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    e, m = eng.estimate_pose_batch(db)
z.record(); torch.cuda.synchronize()
ms = a.elapsed_time(z) / 5
got = eng.edges_to_numpy(e)
print("%-40s %.3f ms  %.0f edges/s  hyps %.1f  E-checksum %.9f" % (os.path.basename(sys.argv[1]), ms, P / ms * 1e3, got["iters"].mean(), float(np.abs(got["E"]).sum())))
