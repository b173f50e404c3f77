"""End-to-end pose-graph build on a synthetic scene graph (surrogate of BASELINE configs 3/4: 1DSfM data is
absent): candidate pairs -> per-edge estimatePose (GPU) -> edge table -> L1/IRLS rotation averaging (GPU)
-> global rotations vs ground truth; per-edge results vs the CPU oracle."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as O
from pyposegraphbuilder import synthetic as S

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import rotavg_oracle as RO  # noqa: E402


def test_scene_graph_generator_is_consistent():
    g = S.make_scene_graph(24, k=4, seed=1)
    b = g["batch"]
    assert len(g["pairs"]) == len(g["sizes"]) and int(b["offsets"][-1]) == g["sizes"].sum()
    for e, (i, j) in enumerate(g["pairs"][:6]):
        np.testing.assert_allclose(b["R"][e], g["R_gt"][j] @ g["R_gt"][i].T, atol=1e-12)
        a, z = int(b["offsets"][e]), int(b["offsets"][e + 1])
        if g["wrong"][e]:
            continue
        E = O.ref_essential_from_pose(b["R"][e], b["t"][e])
        c = np.stack([b["x1"][a:z], b["y1"][a:z], b["x2"][a:z], b["y2"][a:z]], 1).astype(np.float64)
        s2 = np.array([O.ref_sampson_sq(ci, E) for ci in c])
        assert (s2[b["inlier"][a:z]] < (3e-3) ** 2).mean() > 0.99  # inliers satisfy the ground-truth geometry


@pytest.mark.gpu
@pytest.mark.parametrize("V,k", [(60, 6), (340, 20)])
def test_pose_graph_build_and_rotation_averaging(V, k):
    from pyposegraphbuilder import Engine
    g = S.make_scene_graph(V, k=k, seed=3)
    b = g["batch"]
    eng = Engine()
    db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, seed=17)
    edges, masks = eng.estimate_pose_batch(db)
    got = eng.edges_to_numpy(edges)
    # per-edge parity with the oracle on the whole graph
    exp, emasks = O.estimate_pose_batch(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4,
                                        O.default_params(), 17)
    assert np.array_equal(masks.cpu().numpy(), emasks)
    assert np.array_equal(got["E"], exp["E"]) and list(got["status"]) == list(exp["status"])
    ok = got["status"] == 1
    good = ok & ~g["wrong"]
    errs = np.array([S.rot_err_deg(got["R"][e].reshape(3, 3), b["R"][e]) for e in np.nonzero(good)[0]])
    assert S.auc_at(errs) > 0.9 and good.sum() > 0.9 * (~g["wrong"]).sum()
    # pose graph -> rotation averaging; weight = inlier ratio (pose_graph_builder.h:645-654)
    src, dst = g["pairs"][ok, 0], g["pairs"][ok, 1]
    w = got["n_inl"][ok] / g["sizes"][ok]
    R, iters = eng.rotation_average(src, dst, got["R"][ok].reshape(-1, 3, 3), w, V)
    err = RO.align_error_deg(R, g["R_gt"])
    assert err.mean() < 0.5 and np.median(err) < 0.4, (err.mean(), err.max())
    Ro, _ = RO.rotation_average(V, src, dst, got["R"][ok].reshape(-1, 3, 3), w)
    d = np.einsum("kij,kmj->kim", R, Ro)
    assert np.arccos(np.clip((np.trace(d, axis1=1, axis2=2) - 1) / 2, -1, 1)).max() < 1e-5
    eng.close()
