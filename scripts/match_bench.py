"""Descriptor-matching throughput: P pairs of K x K SIFT-sized sets; reports pairs/s and the f32-MFMA roofline
fraction (2*K^2*128 flop per pair; peak 157.3 TFLOP/s, MI355X_MICROARCH.md)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-graph-initialization_amd"))
from pyposegraphbuilder import Engine, synthetic as S


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
    n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    rng = np.random.default_rng(0)
    eng = Engine()
    base, _, _ = S.make_descriptors(rng, K, 2, overlap=0.0)
    sets = []
    for v in range(n_img):
        keep = rng.random(K) < 0.6
        d = np.abs(base + 0.04 * rng.standard_normal(base.shape).astype(np.float32))
        fresh, _, _ = S.make_descriptors(rng, K, 2, overlap=0.0)
        d[~keep] = fresh[~keep]
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        sets.append(d.astype(np.float32)[rng.permutation(K)])
    images = [eng.prepare_descriptors(d) for d in sets]
    pairs = [(i, j) for i in range(n_img) for j in range(n_img) if i != j]
    for P in (1, len(pairs)):
        sel = pairs[:P]
        out = eng.match_descriptors_batch(images, sel, raw=True)
        torch.cuda.synchronize()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            out = eng.match_descriptors_batch(images, sel, raw=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        flop = 2.0 * K * K * 128 * P          # useful flop: the padding to 256 rows is not counted
        import ctypes as C
        f, b = C.c_uint64(0), C.c_uint64(0)
        if hasattr(eng._lib, "pgi_internal_match_flagged"):
            eng._lib.pgi_internal_match_flagged.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
            eng._lib.pgi_internal_match_flagged(eng._ctx, C.byref(f), C.byref(b))
        print("   rows re-scanned exactly: %d forward, %d backward (of %d)" % (f.value, b.value, P * K))
        print("K=%d pairs=%d: %.3f ms  %.1f pairs/s  %.1f f32-equivalent TFLOP/s (%.1f%% of 157.3; >100%% possible on the screened path)  mean matches %.0f" %
              (K, P, dt * 1e3, P / dt, flop / dt / 1e12, 100 * flop / dt / 157.3e12, out[3][:P].float().mean().item()))
    t0 = time.perf_counter()
    for _ in range(20):
        eng.prepare_descriptors(images and torch.as_tensor(sets[0]).cuda())
    torch.cuda.synchronize()
    print("prepare (incl. H2D of 4 MB): %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))


if __name__ == "__main__":
    main()
