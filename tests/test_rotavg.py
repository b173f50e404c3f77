"""Rotation averaging: the numpy/scipy oracle on known graphs (CPU) and the HIP solver against it (GPU)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import rotavg_oracle as RO  # noqa: E402


def test_oracle_exact_on_noise_free_graph():
    src, dst, Rrel, w, Rgt, _ = RO.make_graph(40, 5, noise_deg=0.0, outlier_frac=0.0, seed=1)
    R, iters = RO.rotation_average(40, src, dst, Rrel, w)
    assert RO.align_error_deg(R, Rgt).max() < 1e-4  # arccos resolution near 1
    assert iters <= 2


def test_oracle_rejects_outlier_edges():
    src, dst, Rrel, w, Rgt, out = RO.make_graph(50, 6, noise_deg=1.0, outlier_frac=0.15, seed=3)
    R0, _ = RO.spanning_forest_init(50, src, dst, Rrel, w)
    R, iters = RO.rotation_average(50, src, dst, Rrel, w)
    e0, e = RO.align_error_deg(R0, Rgt), RO.align_error_deg(R, Rgt)
    assert e.mean() < 0.6 and e.max() < 1.5 and e.mean() < 0.5 * e0.mean()
    # consistent with every inlier edge to ~noise level
    res = np.einsum("eji,ejk,ekl->eil", R[dst], Rrel, R[src])
    ang = np.degrees(np.arccos(np.clip((np.trace(res, axis1=1, axis2=2) - 1) / 2, -1, 1)))
    assert np.median(ang[~out]) < 1.5 and np.median(ang[out]) > 20


def test_oracle_disconnected_components_and_gauge():
    src, dst, Rrel, w, Rgt, _ = RO.make_graph(30, 4, noise_deg=0.0, outlier_frac=0.0, seed=5, components=3)
    R, _ = RO.rotation_average(30, src, dst, Rrel, w)
    for c in range(3):  # each component has its own gauge: its smallest view is the identity
        np.testing.assert_allclose(R[c], np.eye(3), atol=1e-12)
        idx = np.arange(c, 30, 3)
        G = Rgt[c].T @ R[c]
        for k in idx:
            np.testing.assert_allclose(Rgt[k] @ G, R[k], atol=1e-8)


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v4_rotavg.npz")
GOLDEN_GRAPHS = ("exact", "noisy_outliers", "three_components", "sparse_ring")


def _golden(name):
    g = np.load(GOLDEN)
    return {k: g[name + "/" + k] for k in ("V", "src", "dst", "Rrel", "weight", "R_gt", "R", "iters")}


def _angles(Ra, Rb):
    d = np.einsum("kij,kmj->kim", Ra, Rb)
    return np.arccos(np.clip((np.trace(d, axis1=1, axis2=2) - 1) / 2, -1, 1))


@pytest.mark.parametrize("name", GOLDEN_GRAPHS)
def test_oracle_reproduces_the_committed_rotation_averaging_fixture(name):
    """four small graphs with their expected absolute rotations as committed data
    (tests/golden/make_golden_rotavg.py).  The oracle of today must still produce them."""
    g = _golden(name)
    assert list(np.load(GOLDEN)["names"]) == list(GOLDEN_GRAPHS)
    R, iters = RO.rotation_average(int(g["V"]), g["src"], g["dst"], g["Rrel"], g["weight"])
    assert iters == int(g["iters"])
    assert np.abs(R - g["R"]).max() < 1e-10   # (element-wise: arccos resolves nothing below 3e-8 rad near the identity)
    if name == "exact":   # and the fixture itself is right where the answer is known in closed form
        assert RO.align_error_deg(g["R"], g["R_gt"]).max() < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("name", GOLDEN_GRAPHS)
def test_hip_rotation_averaging_matches_the_committed_fixture(name):
    from pyposegraphbuilder import Engine
    eng = Engine()
    g = _golden(name)
    R, iters = eng.rotation_average(g["src"], g["dst"], g["Rrel"], g["weight"], int(g["V"]))
    assert _angles(R, g["R"]).max() < 1e-5, (name, _angles(R, g["R"]).max())   # rad: PCG vs sparse direct solve
    assert abs(iters - int(g["iters"])) <= 1
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("V,k,noise,outl,comps", [(40, 5, 0.0, 0.0, 1), (50, 6, 1.0, 0.15, 1), (340, 20, 1.0, 0.2, 1),
                                                  (30, 4, 0.5, 0.1, 3), (1500, 12, 2.0, 0.25, 1)])
def test_hip_rotation_averaging_matches_oracle(V, k, noise, outl, comps):
    from pyposegraphbuilder import Engine
    eng = Engine()
    src, dst, Rrel, w, Rgt, _ = RO.make_graph(V, k, noise, outl, seed=7, components=comps)
    R, iters = eng.rotation_average(src, dst, Rrel, w, V)
    Ro, iters_o = RO.rotation_average(V, src, dst, Rrel, w)
    d = np.einsum("kij,kmj->kim", R, Ro)
    ang = np.arccos(np.clip((np.trace(d, axis1=1, axis2=2) - 1) / 2, -1, 1))
    assert ang.max() < 1e-5, (ang.max(), iters, iters_o)  # rad; PCG vs sparse direct solve, ocml vs libm
    assert abs(iters - iters_o) <= 1
    np.testing.assert_allclose(np.einsum("kij,kmj->kim", R, R), np.tile(np.eye(3), (V, 1, 1)), atol=1e-9)
    if comps == 1:
        assert RO.align_error_deg(R, Rgt).mean() < max(0.05, noise)
    eng.close()


def sequence_graph(V, reach, noise_deg, outlier_frac, seed, components=1):
    """An image-sequence view graph: view i sees views i+1 .. i+reach (per component), like a video or a walk-through.
    Its Laplacian is banded: Jacobi-preconditioned CG needs ~(V / reach) iterations times a large constant."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(seed)
    Rgt = Rotation.random(V, random_state=seed).as_matrix()
    src, dst = [], []
    for c in range(components):
        idx = np.arange(c, V, components)
        for a in range(len(idx)):
            for s_ in range(1, reach + 1):
                if a + s_ < len(idx):
                    i, j = (idx[a], idx[a + s_]) if rng.random() < 0.5 else (idx[a + s_], idx[a])
                    src.append(i); dst.append(j)
    src, dst = np.array(src), np.array(dst)
    Rrel = np.einsum("eij,ekj->eik", Rgt[dst], Rgt[src])
    Rrel = np.einsum("eij,ejk->eik", Rotation.from_rotvec(rng.standard_normal((len(src), 3)) * np.deg2rad(noise_deg) / np.sqrt(3)).as_matrix(), Rrel)
    out = rng.random(len(src)) < outlier_frac
    Rrel[out] = Rotation.random(int(out.sum()), random_state=seed + 1).as_matrix()
    w = np.where(out, rng.uniform(0.1, 0.4, len(src)), rng.uniform(0.4, 1.0, len(src)))
    return src, dst, Rrel, w, Rgt


@pytest.mark.gpu
@pytest.mark.parametrize("V,reach,comps", [(900, 3, 1), (3000, 4, 1), (5000, 3, 2), (6144, 5, 1)])
def test_hip_rotation_averaging_on_sequence_graphs(V, reach, comps, capfd, monkeypatch):
    """Sparse, banded view graphs: the Jacobi-preconditioned solve runs into its cap and the solver must switch to the
    spanning-tree-preconditioned kernel (prefix-sum tree solves) -- same fixed point as the oracle's exact solves, in about
    as many outer iterations, instead of a hundred truncated steps."""
    from pyposegraphbuilder import Engine
    monkeypatch.setenv("PGI_ROTAVG_TRACE", "1")
    eng = Engine()
    src, dst, Rrel, w, Rgt = sequence_graph(V, reach, 1.0, 0.05, seed=13, components=comps)
    R, iters = eng.rotation_average(src, dst, Rrel, w, V)
    trace = capfd.readouterr().err
    Ro, iters_o = RO.rotation_average(V, src, dst, Rrel, w)
    d = np.einsum("kij,kmj->kim", R, Ro)
    ang = np.arccos(np.clip((np.trace(d, axis1=1, axis2=2) - 1) / 2, -1, 1))
    print("V %d: %d outer iterations (oracle %d), max diff %.2e rad" % (V, iters, iters_o, ang.max()))
    assert "): -" in trace                      # tree-preconditioned iterations are traced with a minus sign
    assert ang.max() < 1e-5, (ang.max(), iters, iters_o)
    assert abs(iters - iters_o) <= 1 and iters < 40
    R2, iters2 = eng.rotation_average(src, dst, Rrel, w, V)   # reproducible to the bit
    assert iters2 == iters and np.array_equal(R, R2)
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("V,reach,comps", [(3000, 12, 1), (5000, 20, 1), (4000, 10, 2)])
def test_hip_rotation_averaging_on_dense_band_graphs(V, reach, comps, capfd, monkeypatch):
    """Band graphs too dense for the tree path (more than 8 edges per view) on which every Jacobi solve would run into the cap
    -- config 4's shape: the breadth-first walk from the gauge view is deep, so the solver takes the two-level preconditioner
    (aggregates of neighbouring views, coarse matrix inverted on the device per outer step) and SOLVES its systems, like the
    oracle's sparse direct solves: same fixed point, as many outer iterations, the same bits twice."""
    from pyposegraphbuilder import Engine
    monkeypatch.setenv("PGI_ROTAVG_TRACE", "1")
    eng = Engine()
    src, dst, Rrel, w, Rgt = sequence_graph(V, reach, 1.0, 0.05, seed=17, components=comps)
    assert len(src) > 8 * V
    R, iters = eng.rotation_average(src, dst, Rrel, w, V)
    trace = capfd.readouterr().err
    Ro, iters_o = RO.rotation_average(V, src, dst, Rrel, w)
    d = np.einsum("kij,kmj->kim", R, Ro)
    ang = np.arccos(np.clip((np.trace(d, axis1=1, axis2=2) - 1) / 2, -1, 1))
    two = [int(ln.split("):")[1].split()[0]) for ln in trace.splitlines() if "(two-level)" in ln]
    print("V %d reach %d: %d outer iterations (oracle %d), max diff %.2e rad, two-level iterations %s" % (V, reach, iters, iters_o, ang.max(), two))
    assert len(two) == iters and max(two[1:]) < 150   # (the first L1 step may still run into the cap of 200)
    assert ang.max() < 1e-5, (ang.max(), iters, iters_o)
    assert abs(iters - iters_o) <= 1 and iters < 40
    R2, iters2 = eng.rotation_average(src, dst, Rrel, w, V)
    assert iters2 == iters and np.array_equal(R, R2)
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("V,k,noise,outl,comps", [(50, 6, 1.0, 0.15, 1), (340, 20, 1.0, 0.2, 1), (30, 4, 0.5, 0.1, 3), (1500, 12, 2.0, 0.25, 1),
                                                  (5000, 20, 1.0, 0.15, 1)])
def test_two_level_solver_forced_on_well_conditioned_graphs(V, k, noise, outl, comps, monkeypatch):
    """PGI_ROTAVG_TWO_LEVEL=2 sends every solve through the two-level kernels, also where Jacobi would do (random graphs, tiny
    graphs with two or three aggregates, several components): the same answers as the oracle."""
    from pyposegraphbuilder import Engine
    monkeypatch.setenv("PGI_ROTAVG_TWO_LEVEL", "2")
    eng = Engine()
    src, dst, Rrel, w, Rgt, _ = RO.make_graph(V, k, noise, outl, seed=7, components=comps)
    R, iters = eng.rotation_average(src, dst, Rrel, w, V)
    if V <= 1500:
        Ro, iters_o = RO.rotation_average(V, src, dst, Rrel, w)
        d = np.einsum("kij,kmj->kim", R, Ro)
        ang = np.arccos(np.clip((np.trace(d, axis1=1, axis2=2) - 1) / 2, -1, 1))
        assert ang.max() < 1e-5 and abs(iters - iters_o) <= 1, (ang.max(), iters, iters_o)
    if comps == 1:
        assert RO.align_error_deg(R, Rgt).mean() < max(0.5, noise)
    R2, iters2 = eng.rotation_average(src, dst, Rrel, w, V)
    assert iters2 == iters and np.array_equal(R, R2)
    eng.close()


@pytest.mark.gpu
def test_dense_graph_with_a_capped_first_step_stays_on_jacobi(capfd, monkeypatch):
    """5000 views x ~21 edges per view: the first L1 solve runs into the 200-iteration cap, but the spanning forest is a
    far worse preconditioner there (1 380 iterations against 19) -- the solver must not switch.  (Inner tolerance pinned to the
    1e-10 of rounds 1-4: at the default 1e-4 that first solve stops before the cap and the rule is not exercised.)"""
    from pyposegraphbuilder import Engine
    monkeypatch.setenv("PGI_ROTAVG_TRACE", "1")
    monkeypatch.setenv("PGI_ROTAVG_CG_TOL", "1e-10")
    eng = Engine()
    src, dst, Rrel, w, Rgt, _ = RO.make_graph(5000, 20, noise_deg=1.0, outlier_frac=0.15, seed=2)
    R, iters = eng.rotation_average(src, dst, Rrel, w, 5000)
    trace = capfd.readouterr().err
    assert "(L1): 200 PCG" in trace and "): -" not in trace and "two-level" not in trace
    assert iters <= 12 and RO.align_error_deg(R, Rgt).mean() < 0.5
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["random", "band", "sequence", "chain_with_weak_closures", "weak_bridges"])
def test_inner_tolerance_does_not_move_the_fixed_point(kind, monkeypatch):
    """The inner solves stop at a relative 1e-4 by default (PGI_ROTAVG_CG_TOL overrides): the same outer iteration count and
    the same rotations as with solves to 1e-10, to well under the distance either keeps from the oracle's direct solves.
    ADVICE r5 (low): also on ILL-CONDITIONED graphs -- a 2000-view chain closed by a handful of weak long-range edges, and two
    dense clusters joined by three edges of weight 1e-3 -- where a loose solve could leave the slow modes (the drift along the
    chain, the relative rotation of the clusters) unconverged."""
    from pyposegraphbuilder import Engine
    from scipy.spatial.transform import Rotation
    if kind == "random":
        V = 1500
        src, dst, Rrel, w, _, _ = RO.make_graph(V, 12, noise_deg=2.0, outlier_frac=0.25, seed=5)
    elif kind == "chain_with_weak_closures":
        V = 2000
        src, dst, Rrel, w, Rgt = sequence_graph(V, 1, 0.5, 0.0, seed=23)
        rng = np.random.default_rng(23)
        a = rng.integers(0, V // 2, 6)
        b = a + rng.integers(V // 3, V // 2, 6)
        noise = Rotation.from_rotvec(rng.standard_normal((6, 3)) * np.deg2rad(0.5) / np.sqrt(3)).as_matrix()
        src, dst = np.concatenate([src, a]), np.concatenate([dst, b])
        Rrel = np.concatenate([Rrel, np.einsum("eij,ejk->eik", noise, np.einsum("eij,ekj->eik", Rgt[b], Rgt[a]))])
        w = np.concatenate([w, np.full(6, 0.02)])
    elif kind == "weak_bridges":
        V = 1600
        s1, d1, R1, w1, g1, _ = RO.make_graph(V // 2, 10, noise_deg=1.0, outlier_frac=0.1, seed=7)
        s2, d2, R2, w2, g2, _ = RO.make_graph(V // 2, 10, noise_deg=1.0, outlier_frac=0.1, seed=8)
        a, b = np.array([3, 400, 777]), np.array([5, 123, 650]) + V // 2
        Rgt = np.concatenate([g1, g2])
        src, dst = np.concatenate([s1, s2 + V // 2, a]), np.concatenate([d1, d2 + V // 2, b])
        Rrel = np.concatenate([R1, R2, np.einsum("eij,ekj->eik", Rgt[b], Rgt[a])])
        w = np.concatenate([w1, w2, np.full(3, 1e-3)])
    else:
        V = 3000
        src, dst, Rrel, w, _ = sequence_graph(V, 12 if kind == "band" else 4, 1.0, 0.05, seed=19)
    got = {}
    for tol in ("", "1e-10"):
        if tol:
            monkeypatch.setenv("PGI_ROTAVG_CG_TOL", tol)
        eng = Engine()
        got[tol] = eng.rotation_average(src, dst, Rrel, w, V)
        eng.close()
    assert got[""][1] == got["1e-10"][1], (got[""][1], got["1e-10"][1])
    assert np.abs(got[""][0] - got["1e-10"][0]).max() < 1e-6


@pytest.mark.gpu
def test_hip_rotation_averaging_edge_cases():
    from pyposegraphbuilder import Engine, PgiError
    eng = Engine()
    R, it = eng.rotation_average(np.zeros(0, int), np.zeros(0, int), np.zeros((0, 3, 3)), np.zeros(0), 4)
    np.testing.assert_array_equal(R, np.tile(np.eye(3), (4, 1, 1)))
    with pytest.raises(PgiError):
        eng.rotation_average([0], [7], [np.eye(3)], [1.0], 3)
    eng.close()
