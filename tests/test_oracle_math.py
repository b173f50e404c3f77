"""Pins the CPU oracle against INDEPENDENT numpy/scipy arithmetic.

The reference has no tests and cannot be compiled here, so each restated function is
checked against a different implementation of the same mathematics (numpy.linalg,
numpy.roots, scipy Rotation), never against itself.
"""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

import oracle_lib as O
from pyposegraphbuilder import synthetic as S

rng = np.random.default_rng(7)


def skew(t):
    return np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])


def rand_pose(r=rng):
    R = Rotation.random(random_state=r.integers(1 << 31)).as_matrix()
    t = r.standard_normal(3)
    return R, t / np.linalg.norm(t)


def np_sampson_sq(c, E):
    p1 = np.array([c[0], c[1], 1.0])
    p2 = np.array([c[2], c[3], 1.0])
    Ep1, Etp2 = E @ p1, E.T @ p2
    return (p2 @ E @ p1) ** 2 / (Ep1[0] ** 2 + Ep1[1] ** 2 + Etp2[0] ** 2 + Etp2[1] ** 2)


def test_ref_sampson_matches_textbook():
    # graph_traversal.h:86-116 == standard Sampson distance with r = p2^T E p1
    for _ in range(64):
        E = rng.standard_normal((3, 3))
        c = rng.uniform(-0.5, 0.5, 4)
        assert O.ref_sampson_sq(c, E) == pytest.approx(np_sampson_sq(c, E), rel=1e-12)


def test_ref_essential_from_pose():
    for _ in range(16):
        R, t = rand_pose()
        np.testing.assert_allclose(O.ref_essential_from_pose(R, t), skew(t) @ R, atol=1e-15)


def test_ref_get_inliers_quirk_and_pose_tester():
    d = S.make_pair(3, 400)
    corr = np.stack([d["x1"], d["y1"], d["x2"], d["y2"]], 1).astype(np.float64)
    E = skew(d["t"]) @ d["R"]
    s2 = np.array([np_sampson_sq(c, E) for c in corr])
    thr = 7.5e-4
    # quirk (graph_traversal.h:164): squared residual vs UN-squared threshold
    np.testing.assert_array_equal(O.ref_get_inliers(corr, E, 1.5 * thr), np.nonzero(s2 < 1.5 * thr)[0])
    assert len(O.ref_get_inliers(corr, E, 1.5 * thr)) > d["inlier"].sum()  # far too permissive
    # tester (graph_traversal.h:194-233): squared threshold, stops at the 5th inlier
    ok, n = O.ref_pose_test(corr, d["R"], d["t"], 1.5 * thr, 5)
    assert ok and n == 5
    ok, n = O.ref_pose_test(corr, d["R"], d["t"], 1.5 * thr, 10 ** 6)
    assert (not ok) and n == int((s2 < (1.5 * thr) ** 2).sum())
    Rb, tb = rand_pose()
    ok, n = O.ref_pose_test(corr, Rb, tb, 1.5 * thr, 50)
    assert not ok


def test_ref_chain_pose():
    def T(R, t):
        M = np.eye(4)
        M[:3, :3], M[:3, 3] = R, t
        return M
    R, t = np.eye(3), np.zeros(3)
    M = np.eye(4)
    for k in range(5):
        Re, te = rand_pose()
        inv = bool(k % 2)
        R, t = O.ref_chain_pose(Re, te, inv, R, t)
        M = (np.linalg.inv(T(Re, te)) if inv else T(Re, te)) @ M
    np.testing.assert_allclose(T(R, t), M, atol=1e-13)


def test_ref_normalize_corr():
    ks = rng.uniform(0, 1000, (20, 2)).astype(np.float32)
    kd = rng.uniform(0, 800, (30, 2)).astype(np.float32)
    ms, md = rng.integers(0, 20, 12), rng.integers(0, 30, 12)
    cs, cd = (900.0, 1000.0, 700.0), (1100.0, 800.0, 600.0)
    c, thr = O.ref_normalize_corr(ks, kd, ms, md, cs, cd, False, 0.4)
    exp = np.concatenate([(ks[ms] - [500, 350]) / 900.0, (kd[md] - [400, 300]) / 1100.0], 1)
    np.testing.assert_allclose(c, exp, rtol=1e-14)
    assert thr == pytest.approx(0.4 / 1000.0)
    # pose_graph_builder.h:908-912: destination normalised with SOURCE intrinsics
    c, thr = O.ref_normalize_corr(ks, kd, ms, md, cs, cd, True, 0.4)
    exp = np.concatenate([(ks[ms] - [500, 350]) / 900.0, (kd[md] - [500, 350]) / 900.0], 1)
    np.testing.assert_allclose(c, exp, rtol=1e-14)
    assert thr == pytest.approx(0.4 / 900.0)


MONO = [(3, 0, 0), (0, 3, 0), (2, 1, 0), (1, 2, 0), (2, 0, 1), (2, 0, 0), (0, 2, 1), (0, 2, 0),
        (1, 1, 1), (1, 1, 0), (1, 0, 2), (1, 0, 1), (1, 0, 0), (0, 1, 2), (0, 1, 1), (0, 1, 0),
        (0, 0, 3), (0, 0, 2), (0, 0, 1), (0, 0, 0)]


def test_constraint_matrix_evaluates_the_cubic_constraints():
    basis = np.linalg.qr(rng.standard_normal((9, 4)))[0].T
    _, dbg = O.backend(basis)
    cons = np.array(dbg.cons).reshape(10, 20)
    for _ in range(8):
        x, y, z = rng.standard_normal(3)
        mono = np.array([x ** a * y ** b * z ** c for a, b, c in MONO])
        E = (x * basis[0] + y * basis[1] + z * basis[2] + basis[3]).reshape(3, 3)
        exp = np.concatenate([[np.linalg.det(E)],
                              (E @ E.T @ E - 0.5 * np.trace(E @ E.T) * E).ravel()])
        np.testing.assert_allclose(cons @ mono, exp, atol=1e-12)


def test_polynomial_roots_match_numpy_roots():
    missed = total = 0
    for k in range(40):
        d = S.make_pair(100 + k, 40, inlier_ratio=1.0, noise_px=0.0)
        pts = np.stack([d["x1"], d["y1"], d["x2"], d["y2"]], 1)[:5]
        _, dbg = O.five_point(pts)
        poly = np.array(dbg.poly)
        r = np.roots(poly[::-1])
        real = np.sort(r[np.abs(r.imag) < 1e-9 * (1 + np.abs(r.real))].real)
        mine = np.array(dbg.roots[:dbg.n_roots])
        for z in mine:  # every oracle root is a true root
            assert np.min(np.abs(real - z)) < 1e-7 * (1 + abs(z))
        total += len(real)
        missed += len(real) - len(mine)
    assert total > 0 and missed <= 0.1 * total  # grid bracketing may miss |z|>64 or close pairs


def test_five_point_contains_ground_truth():
    hits = 0
    for k in range(32):
        d = S.make_pair(200 + k, 40, inlier_ratio=1.0, noise_px=0.0)
        pts = np.stack([d["x1"], d["y1"], d["x2"], d["y2"]], 1)[:5]
        Egt = skew(d["t"]) @ d["R"]
        Egt /= np.linalg.norm(Egt)
        models, _ = O.five_point(pts)
        assert len(models) <= 10
        for m in models:
            E = m.reshape(3, 3).astype(np.float64)
            assert abs(np.linalg.norm(E) - 1) < 1e-6
            assert abs(np.linalg.det(E)) < 1e-6
            np.testing.assert_allclose(2 * E @ E.T @ E - np.trace(E @ E.T) * E, 0, atol=2e-6)
            for p in pts.astype(np.float64):  # passes through the sample
                assert abs(np.array([p[2], p[3], 1]) @ E @ np.array([p[0], p[1], 1])) < 1e-6
        err = [min(np.linalg.norm(m.reshape(3, 3) - Egt), np.linalg.norm(m.reshape(3, 3) + Egt))
               for m in models] or [9]
        hits += min(err) < 1e-4
    assert hits >= 30  # the oriented-constraint prune and grid may drop a rare case


def test_nullspace5_is_orthonormal_null():
    d = S.make_pair(5, 20)
    pts = np.stack([d["x1"], d["y1"], d["x2"], d["y2"]], 1)[:5].astype(np.float64)
    B = O.nullspace5(pts)
    np.testing.assert_allclose(B @ B.T, np.eye(4), atol=1e-12)
    A = np.array([[p[2] * p[0], p[2] * p[1], p[2], p[3] * p[0], p[3] * p[1], p[3], p[0], p[1], 1] for p in pts])
    np.testing.assert_allclose(A @ B.T, 0, atol=1e-12)


def test_normal_matrix_exact_and_order_independent():
    d = S.make_pair(6, 500)
    x1, y1, x2, y2 = d["x1"], d["y1"], d["x2"], d["y2"]
    mask = d["inlier"].astype(np.uint8)
    A = O.normal_matrix(x1, y1, x2, y2, mask)
    a = np.stack([x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, np.ones_like(x1)], 1).astype(np.float64)
    a = a[mask.astype(bool)]
    np.testing.assert_allclose(A, a.T @ a, atol=len(a) * 2.0 ** -34)
    perm = rng.permutation(len(x1))
    A2 = O.normal_matrix(x1[perm], y1[perm], x2[perm], y2[perm], mask[perm])
    assert np.array_equal(A, A2)  # bit-identical: summands pre-rounded to 2^-34


def test_jacobi9_matches_eigh():
    for k in range(8):
        d = S.make_pair(300 + k, 300, noise_px=0.25)
        A = O.normal_matrix(d["x1"], d["y1"], d["x2"], d["y2"], d["inlier"].astype(np.uint8))
        D, V = O.jacobi9(A)
        w = np.linalg.eigvalsh(A)
        np.testing.assert_allclose(np.sort(np.diag(D)), w, atol=1e-11 * w[-1])
        np.testing.assert_allclose(V.T @ V, np.eye(9), atol=1e-12)
        np.testing.assert_allclose(A @ V, V * np.diag(D), atol=1e-9 * w[-1])  # 6 fixed sweeps
        off = D - np.diag(np.diag(D))
        assert np.abs(off).max() < 1e-9 * w[-1]
        B = O.basis_from_eigen(D, V)
        order = np.argsort(np.diag(D), kind="stable")
        np.testing.assert_array_equal(B[3], V[:, order[0]])
        np.testing.assert_array_equal(B[0], V[:, order[3]])


def test_npoint_refit_improves_on_noisy_inliers():
    d = S.make_pair(11, 400, inlier_ratio=1.0, noise_px=0.25)
    models = O.npoint(d["x1"], d["y1"], d["x2"], d["y2"])
    Egt = skew(d["t"]) @ d["R"]
    Egt /= np.linalg.norm(Egt)
    err = min(min(np.linalg.norm(m.reshape(3, 3) - Egt), np.linalg.norm(m.reshape(3, 3) + Egt)) for m in models)
    assert err < 5e-3


def test_svd3_matches_numpy():
    for _ in range(16):
        R, t = rand_pose()
        E = skew(t) @ R + 1e-7 * rng.standard_normal((3, 3))
        U, Sg, V = O.svd3(E)
        np.testing.assert_allclose(U @ np.diag(Sg) @ V.T * np.sign(np.linalg.det(V.T @ np.linalg.pinv(np.diag(Sg)) @ U.T @ E @ V) if False else 1), U @ np.diag(Sg) @ V.T)
        np.testing.assert_allclose(np.sort(Sg)[::-1], np.linalg.svd(E, compute_uv=False), atol=1e-12)
        np.testing.assert_allclose(U.T @ U, np.eye(3), atol=1e-7)
        np.testing.assert_allclose(V.T @ V, np.eye(3), atol=1e-12)
        assert np.linalg.det(U) > 0 and np.linalg.det(V) > 0
        # rank-2 reconstruction up to the sign of the third singular triplet
        E2 = U[:, :2] @ np.diag(Sg[:2]) @ V[:, :2].T
        np.testing.assert_allclose(E2, E, atol=1e-6)


def test_decompose_recovers_ground_truth_pose():
    for k in range(16):
        d = S.make_pair(400 + k, 200, inlier_ratio=1.0, noise_px=0.0)
        E = skew(d["t"]) @ d["R"] * (-1.0) ** k  # sign of E must not matter
        R, t, votes, cand = O.decompose(E, d["x1"], d["y1"], d["x2"], d["y2"], None)
        assert S.rot_err_deg(R, d["R"]) < 1e-4
        assert t @ d["t"] > 1 - 1e-9
        assert votes[cand] == 200 and votes.sum() == 200  # exactly one candidate per point
        assert abs(np.linalg.det(R) - 1) < 1e-9


def test_sample5_distinct_deterministic():
    seen = set()
    for h in range(200):
        idx = O.sample5(1234, 7, h, 50)
        assert len(set(idx.tolist())) == 5 and idx.max() < 50
        seen.add(tuple(idx))
        np.testing.assert_array_equal(idx, O.sample5(1234, 7, h, 50))
    assert len(seen) > 190
    assert O.lib().pgo_mix64(0) == 0xE220A8397B1DCDAF  # splitmix64 known answer


def test_score_levels_and_mask():
    d = S.make_pair(12, 600)
    E = (skew(d["t"]) @ d["R"]).astype(np.float32)
    thr = 7.5e-4
    x1, y1, x2, y2 = (d[k].astype(np.float64) for k in ("x1", "y1", "x2", "y2"))
    c = np.stack([x1, y1, x2, y2], 1)
    s2 = np.array([np_sampson_sq(ci, E.astype(np.float64)) for ci in c])
    lv = [(s2 < (f * thr) ** 2) for f in (0.5, 0.75, 1.0, 1.5)]
    score, n_inl = O.score_model(E, d["x1"], d["y1"], d["x2"], d["y2"], thr)
    # f32 arithmetic may flip borderline points only
    border = sum(int((np.abs(np.sqrt(s2) / (f * thr) - 1) < 1e-3).sum()) for f in (0.5, 0.75, 1.0, 1.5))
    assert abs(score - sum(int(l.sum()) for l in lv)) <= border
    assert abs(n_inl - int(lv[2].sum())) <= border
    m, cnt = O.mask_model(E, d["x1"], d["y1"], d["x2"], d["y2"], np.float32(thr * thr))
    assert cnt == n_inl == m.sum()


@pytest.mark.parametrize("rho,thr", [(0.5, 7.5e-4), (0.7, 7.5e-4), (0.5, 4e-4)])
def test_estimate_pose_statistical_quality(rho, thr):
    b = S.make_batch(range(2000, 2024), 600, inlier_ratio=rho)
    out, masks = O.estimate_pose_batch(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], thr,
                                       O.default_params(), 99)
    errs = [S.rot_err_deg(out["R"][i].reshape(3, 3), b["R"][i]) if out["status"][i] == 1 else np.inf
            for i in range(24)]
    assert S.auc_at(errs) > 0.93
    assert all(abs(out["t"][i] @ b["t"][i]) > 0.99 for i in range(24) if errs[i] < 1)
    # mask agrees with the generator's labels (a few outliers fall on the epipolar band)
    agree = (masks.astype(bool) == b["inlier"]).mean()
    assert agree > 0.9


def test_estimate_pose_guess_path_and_quirk():
    d = S.make_pair(77, 800)
    prm = O.default_params()
    guess = np.concatenate([d["R"].ravel(), d["t"]])
    e, mask = O.estimate_pose(d["x1"], d["y1"], d["x2"], d["y2"], 7.5e-4, guess, prm, 5, 77)
    assert e.status == 1 and e.used_guess == 1 and e.iters == 0
    # quirk threshold 1.5*thr on the SQUARED residual admits far more than the true inliers
    assert e.n_inl == mask.sum() and e.n_inl > d["inlier"].sum()
    assert S.rot_err_deg(np.array(e.R).reshape(3, 3), d["R"]) < 1.0
    prm2 = O.default_params(guess_quirk=0)
    e2, mask2 = O.estimate_pose(d["x1"], d["y1"], d["x2"], d["y2"], 7.5e-4, guess, prm2, 5, 77)
    assert e2.used_guess == 1 and e2.n_inl < e.n_inl
    assert S.rot_err_deg(np.array(e2.R).reshape(3, 3), d["R"]) < 0.2
    # a wrong guess falls through to the robust fit (pose_graph_builder.h:1031)
    Rb, tb = rand_pose(np.random.default_rng(3))
    e3, _ = O.estimate_pose(d["x1"], d["y1"], d["x2"], d["y2"], 7.5e-4, np.concatenate([Rb.ravel(), tb]),
                            prm2, 5, 77)
    assert e3.status == 1 and e3.used_guess == 0 and e3.iters > 0


def test_edge_cases():
    prm = O.default_params()
    z = np.zeros(3, np.float32)
    e, m = O.estimate_pose(z, z, z, z, 7.5e-4, None, prm, 1, 0)
    assert e.status == -2  # fewer than 5 points
    r = np.random.default_rng(0)
    x = [r.uniform(-0.5, 0.5, 60).astype(np.float32) for _ in range(4)]
    e, m = O.estimate_pose(*x, 7.5e-4, None, prm, 1, 0)  # pure outliers
    assert e.status == 0 and e.iters == prm.max_iters + (-prm.max_iters) % prm.round_size
    fx = O.default_params(fixed_budget=64)
    d = S.make_pair(1, 300)
    e, m = O.estimate_pose(d["x1"], d["y1"], d["x2"], d["y2"], 7.5e-4, None, fx, 1, 1)
    assert e.iters == 64 and e.status == 1


def test_auc():
    assert S.auc_at([0, 0, 0, 0]) == 1.0
    assert S.auc_at([np.inf, 10]) == 0.0
    assert S.auc_at([2.5, np.inf]) == pytest.approx(0.25)
