import os, sys, time
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "pose-graph-initialization_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rotavg_oracle as RO
from test_rotavg import sequence_graph
from pyposegraphbuilder import Engine
eng = Engine()
for name, (V, mk) in {"band 20000 x 20": (20000, lambda: sequence_graph(20000, 20, 1.0, 0.05, seed=3)), "band 12000 x 30, 3 components": (12000, lambda: sequence_graph(12000, 30, 1.0, 0.05, seed=4, components=3)),
                      "random 20000 x 20": (20000, lambda: RO.make_graph(20000, 20, noise_deg=1.0, outlier_frac=0.15, seed=2)[:5])}.items():
    src, dst, Rrel, w, Rgt = mk()
    eng.rotation_average(src, dst, Rrel, w, V)
    t0 = time.perf_counter(); R, iters = eng.rotation_average(src, dst, Rrel, w, V); dt = time.perf_counter() - t0
    R2, it2 = eng.rotation_average(src, dst, Rrel, w, V)
    ortho = np.abs(np.einsum("kij,kmj->kim", R, R) - np.eye(3)).max()
    print("%-32s E=%7d: %.1f ms, %d outer iterations, same bits twice %s, |RR^T - I| %.1e" % (name, len(src), 1e3 * dt, iters, np.array_equal(R, R2) and it2 == iters, ortho), flush=True)
