import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo/pose-graph-initialization_amd")
from pyposegraphbuilder import _lib as L
if len(sys.argv) > 1: L.LIB_PATH = os.path.join("/root/repo/pose-graph-initialization_amd", sys.argv[1])
from pyposegraphbuilder import Engine, synthetic as S
K, n_img = 8000, 8
rng = np.random.default_rng(0)
eng = Engine()
sets = []
for v in range(n_img):
    d, _, _ = S.make_descriptors(rng, K, 2, overlap=0.0)
    sets.append(d)
images = [eng.prepare_descriptors(d) for d in sets]
pairs = [(i, j) for i in range(n_img) for j in range(n_img) if i != j]
eng.match_descriptors_batch(images, pairs, raw=True); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): eng.match_descriptors_batch(images, pairs, raw=True)
torch.cuda.synchronize()
print(sys.argv[1:], "%.3f ms per 56 pairs" % ((time.perf_counter() - t0) / 5 * 1e3))
