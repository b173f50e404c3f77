"""Generates tests/golden/golden_v4_rotavg.npz (run in the build container: `python tests/golden/make_golden_rotavg.py`).

SURVEY section 8c(7): "rotation averaging on 4 small graphs" as committed data.  Four view graphs with known rotations --
noise-free, noisy with outlier edges, three disconnected components (one gauge each), and a sparse sequence graph -- as
INPUTS (src, dst, relative rotations, weights) and EXPECTED OUTPUTS (the absolute rotations and the outer-iteration count of
oracle/rotavg_oracle.py: scipy log/exp maps + sparse direct solves), plus the ground truth the graphs were made from.
tests/test_rotavg.py checks the oracle against this file on CPU (so a change of the oracle, of scipy or of numpy that moves
the result shows up as a diff against committed numbers) and the HIP solver against it on the GPU.  Data only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import rotavg_oracle as RO  # noqa: E402

# name: (views, neighbours per view, noise in degrees, outlier-edge fraction, seed, components)
GRAPHS = {
    "exact": (24, 4, 0.0, 0.0, 101, 1),
    "noisy_outliers": (50, 6, 1.0, 0.15, 102, 1),
    "three_components": (30, 4, 0.5, 0.10, 103, 3),
    "sparse_ring": (40, 1, 1.0, 0.0, 104, 1),
}


def main():
    out = {"names": np.array(list(GRAPHS))}
    for name, (V, k, noise, outl, seed, comps) in GRAPHS.items():
        src, dst, Rrel, w, Rgt, bad = RO.make_graph(V, k, noise, outl, seed=seed, components=comps)
        R, iters = RO.rotation_average(V, src, dst, Rrel, w)
        out.update({name + "/V": np.int64(V), name + "/src": src.astype(np.int64), name + "/dst": dst.astype(np.int64),
                    name + "/Rrel": Rrel, name + "/weight": w, name + "/R_gt": Rgt, name + "/outlier": bad,
                    name + "/R": R, name + "/iters": np.int64(iters)})
        print("%-18s V %3d E %4d iters %2d  mean error vs ground truth %.4f deg" % (
            name, V, len(src), iters, RO.align_error_deg(R, Rgt).mean() if comps == 1 else float("nan")))
    np.savez_compressed(os.path.join(HERE, "golden_v4_rotavg.npz"), **out)


if __name__ == "__main__":
    main()
