"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.

Bars (BASELINE.json:north_star): identical inlier masks under fixed seeds; R within 1e-4 rad;
t-direction cosine within 1e-3.  The specification is deterministic, so the tests demand more:
bit-identical E, masks, counts and iteration numbers.
"""
import os

import numpy as np
import pytest

import oracle_lib as O
from pyposegraphbuilder import synthetic as S

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v1.npz"))
R_TOL_RAD = 1e-4
T_COS_TOL = 1e-3


@pytest.fixture(scope="module")
def eng():
    from pyposegraphbuilder import Engine
    e = Engine()
    yield e
    e.close()


def rot_angle(Ra, Rb):
    c = (np.trace(Ra.reshape(3, 3) @ Rb.reshape(3, 3).T) - 1) / 2
    return np.arccos(np.clip(c, -1, 1))


def assert_edges_match(got, exp, exact=True):
    assert list(got["status"]) == list(exp["status"])
    for k in ("n_inl", "score", "iters", "lo_runs", "used_guess"):
        assert list(got[k]) == list(exp[k]), k
    ok = exp["status"] == 1
    for i in np.nonzero(ok)[0]:
        assert rot_angle(got["R"][i], exp["R"][i]) <= R_TOL_RAD
        assert got["t"][i] @ exp["t"][i] >= 1 - T_COS_TOL
    if exact:
        assert np.array_equal(got["E"], exp["E"])
        assert list(got["cand"]) == list(exp["cand"]) and list(got["votes"]) == list(exp["votes"])
        np.testing.assert_allclose(got["R"], exp["R"], atol=1e-12)
        np.testing.assert_allclose(got["t"], exp["t"], atol=1e-12)


def test_ieee_division_sqrt_are_correctly_rounded(eng):
    """The bit-exact contract needs IEEE f64 /, sqrt, fma on the device: the 5-point debug taps
    (orthonormal basis = division + sqrt chains) must equal the oracle to the last bit."""
    pts = G["fp_pts"]
    _, _, dbg = eng.five_point_batch(pts, debug=True)
    for k in range(len(pts)):
        assert np.array_equal(dbg[k, :36].reshape(4, 9), O.nullspace5(pts[k])), k


STAGES = [("basis", 0, 36), ("cons", 36, 236), ("red", 236, 336), ("poly", 336, 347), ("roots", 347, 357),
          ("nroots", 357, 358)]


def test_five_point_stage_by_stage(eng):
    pts = np.concatenate([G["fp_pts"], np.stack([np.stack([d[k] for k in ("x1", "y1", "x2", "y2")], 1)[:5]
                          for d in (S.make_pair(8000 + i, 40) for i in range(96))])])
    models, counts, dbg = eng.five_point_batch(pts, debug=True)
    for k in range(len(pts)):
        om, od = O.five_point(pts[k])
        exp = dict(basis=O.nullspace5(pts[k]).ravel(), cons=np.array(od.cons), red=np.array(od.red),
                   poly=np.array(od.poly), roots=np.array(od.roots), nroots=np.array([float(od.n_roots)]))
        for name, a, b in STAGES:
            got = dbg[k, a:b]
            assert np.array_equal(got, exp[name]), "sample %d stage %s max|d|=%g" % (
                k, name, np.nanmax(np.abs(got - exp[name])))
        assert counts[k] == len(om)
        assert np.array_equal(models[k, :counts[k]], om)


def test_five_point_golden(eng):
    models, counts, _ = eng.five_point_batch(G["fp_pts"])
    assert list(counts) == list(G["fp_counts"])
    for k, c in enumerate(counts):
        assert np.array_equal(models[k, :c], G["fp_models"][k, :c])


def test_score_pose_batch_matches_oracle(eng):
    sizes = [5, 63, 64, 65, 257, 1000, 2000, 1, 3, 130]
    b = S.make_batch(range(100, 110), sizes)
    E = np.stack([O.ref_essential_from_pose(b["R"][i], b["t"][i]).ravel() for i in range(10)])
    thr = 7.5e-4
    db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], thr)
    for tau2 in (thr * thr, 1.5 * thr, (1.5 * thr) ** 2):
        counts, masks = eng.score_pose_batch(db, E, tau2)
        counts, masks = counts.cpu().numpy(), masks.cpu().numpy()
        for i in range(10):
            a, z = int(b["offsets"][i]), int(b["offsets"][i + 1])
            En = O.model_from_essential(E[i])  # the kernel's own normalisation: f64 fma chain, then rounded to f32
            m, c = O.mask_model(En, b["x1"][a:z], b["y1"][a:z], b["x2"][a:z], b["y2"][a:z], np.float32(tau2))
            assert int(counts[i]) == c and np.array_equal(masks[a:z], m)   # exact: mask for mask
            assert counts[i] == masks[a:z].sum()


def test_score_pose_batch_accepts_unaligned_slices(eng):
    """The SoA arrays may be slices of larger tensors: streams whose 16-byte phase differs between the four arrays (or a
    mask base that is not 4-byte aligned) must take the scalar path and give the same masks as aligned inputs."""
    import torch
    b = S.make_batch(range(140, 146), [700, 64, 5, 1999, 130, 1024])
    E = np.stack([O.ref_essential_from_pose(b["R"][i], b["t"][i]).ravel() for i in range(6)])
    thr = 7.5e-4
    db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], thr)
    ref_counts, ref_masks = eng.score_pose_batch(db, E, thr * thr)
    rows = db["x1"].numel()
    for shifts in ((1, 0, 0, 0), (0, 1, 2, 3), (3, 3, 3, 3), (2, 2, 1, 1)):
        d2 = dict(db)
        for key, sh in zip(("x1", "y1", "x2", "y2"), shifts):
            big = torch.zeros(rows + 8, dtype=torch.float32, device=eng.device)
            big[sh:sh + rows] = db[key]
            d2[key] = big[sh:sh + rows]
        counts, masks = eng.score_pose_batch(d2, E, thr * thr)
        assert torch.equal(counts, ref_counts) and torch.equal(masks, ref_masks), shifts


def test_score_pose_f64_is_bit_identical_to_reference_formula(eng):
    b = S.make_batch(range(120, 126), [50, 64, 100, 333, 1, 2000])
    corr = np.stack([b["x1"], b["y1"], b["x2"], b["y2"]], 1).astype(np.float64)
    E = np.stack([O.ref_essential_from_pose(b["R"][i], b["t"][i]).ravel() for i in range(6)])
    thr = 7.5e-4
    for tau2 in (1.5 * thr, (1.5 * thr) ** 2):  # graph_traversal.h:164 quirk, :184 tester
        counts, masks = eng.score_pose_f64(corr, b["offsets"], E, tau2)
        for i in range(6):
            a, z = int(b["offsets"][i]), int(b["offsets"][i + 1])
            exp = np.array([O.ref_sampson_sq(corr[j], E[i]) < tau2 for j in range(a, z)], np.uint8)
            assert np.array_equal(masks[a:z], exp)
            assert counts[i] == exp.sum()
        if tau2 == 1.5 * thr:
            idx = O.ref_get_inliers(corr[:50], E[0], 1.5 * thr)
            assert np.array_equal(np.nonzero(masks[:50])[0], idx)


def test_decompose_batch_matches_oracle(eng):
    b = S.make_batch(range(130, 138), [200, 300, 64, 1000, 50, 77, 128, 500], noise_px=0.1)
    db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4)
    E = np.stack([O.ref_essential_from_pose(b["R"][i], b["t"][i]).ravel() * (-1) ** i for i in range(8)])
    import torch
    masks = torch.from_numpy(b["inlier"].astype(np.uint8)).to(eng.device)
    got = eng.edges_to_numpy(eng.decompose_batch(db, E, masks))
    for i in range(8):
        a, z = int(b["offsets"][i]), int(b["offsets"][i + 1])
        R, t, votes, cand = O.decompose(E[i], b["x1"][a:z], b["y1"][a:z], b["x2"][a:z], b["y2"][a:z],
                                        b["inlier"][a:z].astype(np.uint8))
        assert got["cand"][i] == cand and got["votes"][i] == votes[cand]
        np.testing.assert_allclose(got["R"][i].reshape(3, 3), R, atol=1e-13)
        np.testing.assert_allclose(got["t"][i], t, atol=1e-13)
        assert S.rot_err_deg(got["R"][i].reshape(3, 3), b["R"][i]) < 1e-3


@pytest.mark.parametrize("tag,kw", [("", {}), ("_fixed", {"fixed_budget": 96})])
def test_estimate_pose_golden(eng, tag, kw):
    eng.set_params(fixed_budget=kw.get("fixed_budget", 0))
    db = eng.upload(G["ep_x1"], G["ep_y1"], G["ep_x2"], G["ep_y2"], G["ep_offsets"], G["ep_thr"],
                    seed=int(G["ep_seed"]), pair_id_base=9000)
    edges, masks = eng.estimate_pose_batch(db)
    got = eng.edges_to_numpy(edges)
    eng.set_params(fixed_budget=0)
    assert np.array_equal(masks.cpu().numpy(), G["ep_masks" + tag])  # identical inlier masks
    assert_edges_match(got, G["ep_out" + tag])


@pytest.mark.parametrize("tag,fixed", [("", 0), ("_fixed", 96)])
def test_estimate_pose_golden_nister_only_lo(eng, tag, fixed):
    """lo_linear_pct = 0 on the device == the fixture's PRE-CHANGE outputs (tests/golden/golden_v1_nister_lo.npz, taken from
    the repository history): the behaviour before the hybrid refit stays pinned."""
    GN = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v1_nister_lo.npz"))
    eng.set_params(fixed_budget=fixed, lo_linear_pct=0)
    try:
        db = eng.upload(G["ep_x1"], G["ep_y1"], G["ep_x2"], G["ep_y2"], G["ep_offsets"], G["ep_thr"],
                        seed=int(G["ep_seed"]), pair_id_base=9000)
        edges, masks = eng.estimate_pose_batch(db)
        got = eng.edges_to_numpy(edges)
    finally:
        eng.set_params(fixed_budget=0, lo_linear_pct=35)
    assert np.array_equal(masks.cpu().numpy(), GN["ep_masks" + tag])
    assert_edges_match(got, GN["ep_out" + tag])


def test_estimate_pose_guess_golden(eng):
    db = eng.upload(G["ep_x1"], G["ep_y1"], G["ep_x2"], G["ep_y2"], G["ep_offsets"], G["ep_thr"],
                    guesses=G["ep_guesses"], seed=int(G["ep_seed"]), pair_id_base=9000)
    edges, masks = eng.estimate_pose_batch(db)
    assert np.array_equal(masks.cpu().numpy(), G["ep_masks_guess"])
    assert_edges_match(eng.edges_to_numpy(edges), G["ep_out_guess"])


@pytest.mark.parametrize("rho,thr,n", [(0.5, 7.5e-4, 2000), (0.3, 7.5e-4, 700), (0.7, 4e-4, 1200)])
def test_estimate_pose_matches_oracle_live(eng, rho, thr, n):
    ids = np.arange(3000, 3024)
    b = S.make_batch(ids, n, inlier_ratio=rho)
    db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], thr, seed=77, pair_id_base=3000)
    edges, masks = eng.estimate_pose_batch(db)
    exp, emasks = O.estimate_pose_batch(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], thr,
                                        O.default_params(), 77, pair_id_base=3000)
    assert np.array_equal(masks.cpu().numpy(), emasks)
    assert_edges_match(eng.edges_to_numpy(edges), exp)


def test_ragged_and_edge_cases(eng):
    sizes = [0, 1, 4, 5, 6, 49, 50, 64, 65, 127, 128, 129, 4000, 3, 8]
    ids = np.arange(4000, 4000 + len(sizes))
    b = S.make_batch(ids, sizes)
    db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, seed=5, pair_id_base=4000)
    edges, masks = eng.estimate_pose_batch(db)
    exp, emasks = O.estimate_pose_batch(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4,
                                        O.default_params(), 5, pair_id_base=4000)
    got = eng.edges_to_numpy(edges)
    assert list(got["status"][:3]) == [-2, -2, -2]
    assert np.array_equal(masks.cpu().numpy(), emasks)
    assert_edges_match(got, exp)
    # pure outliers: no edge (pose_graph_builder.h:1053-1054)
    r = np.random.default_rng(0)
    x = [r.uniform(-0.5, 0.5, 80).astype(np.float32) for _ in range(4)]
    db = eng.upload(*x, np.array([0, 80]), 7.5e-4, seed=1)
    edges, masks = eng.estimate_pose_batch(db)
    exp, emasks = O.estimate_pose_batch(*x, np.array([0, 80]), 7.5e-4, O.default_params(), 1)
    assert_edges_match(eng.edges_to_numpy(edges), exp)
    assert np.array_equal(masks.cpu().numpy(), emasks)


def test_rows_beyond_lds_capacity_use_the_global_path(eng):
    n = 12000  # > the LDS staging capacity: rows stay in HBM/L2, results unchanged
    b = S.make_batch([5000, 5001], n)
    db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, seed=9, pair_id_base=5000)
    edges, masks = eng.estimate_pose_batch(db)
    exp, emasks = O.estimate_pose_batch(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4,
                                        O.default_params(), 9, pair_id_base=5000)
    assert np.array_equal(masks.cpu().numpy(), emasks)
    assert_edges_match(eng.edges_to_numpy(edges), exp)


def test_single_pair_drop_in(eng):
    d = S.make_pair(6000, 1500)
    corr = np.stack([d["x1"], d["y1"], d["x2"], d["y2"]], 1).astype(np.float64)  # cv::Mat N x 4 CV_64F
    ok, e, mask = eng.estimate_pose(corr, 7.5e-4, seed=3, pair_id=6000)
    exp, emask = O.estimate_pose(d["x1"], d["y1"], d["x2"], d["y2"], 7.5e-4, None, O.default_params(), 3, 6000)
    assert ok and e.status == 1 == exp.status
    assert np.array_equal(mask, emask) and e.n_inl == exp.n_inl
    assert np.array_equal(np.array(e.E), np.array(exp.E))
    assert S.rot_err_deg(np.array(e.R).reshape(3, 3), d["R"]) < 0.5
    # guesses: the last one wins (pose_graph_builder.h:974-1029)
    guesses = np.stack([np.concatenate([np.eye(3).ravel(), [1, 0, 0]]), np.concatenate([d["R"].ravel(), d["t"]])])
    ok, e, mask = eng.estimate_pose(corr, 7.5e-4, guesses=guesses, seed=3, pair_id=6000)
    exp, emask = O.estimate_pose(d["x1"], d["y1"], d["x2"], d["y2"], 7.5e-4, guesses[1], O.default_params(), 3, 6000)
    assert ok and e.used_guess == 1 == exp.used_guess and np.array_equal(mask, emask)


def test_full_size_properties(eng):
    """BASELINE config 2 shape (2000 rows per pair) on 512 pairs: size-independent properties."""
    ids = np.arange(20000, 20512)
    b = S.make_batch(ids, 2000)
    db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, seed=11, pair_id_base=20000)
    edges, masks = eng.estimate_pose_batch(db)
    got = eng.edges_to_numpy(edges)
    m = masks.cpu().numpy()
    # idempotence / determinism: a second launch is bit-identical
    edges2, masks2 = eng.estimate_pose_batch(db)
    assert np.array_equal(edges.cpu().numpy(), edges2.cpu().numpy()) and np.array_equal(m, masks2.cpu().numpy())
    # sharding invariance: the second half alone (pair_id_base shifted) reproduces its edges
    h = 256
    a = int(b["offsets"][h])
    db2 = eng.upload(b["x1"][a:], b["y1"][a:], b["x2"][a:], b["y2"][a:], b["offsets"][h:] - b["offsets"][h],
                     7.5e-4, seed=11, pair_id_base=20000 + h)
    e3, m3 = eng.estimate_pose_batch(db2)
    assert np.array_equal(eng.edges_to_numpy(e3).tobytes(), got[h:].tobytes()) and np.array_equal(m3.cpu().numpy(), m[a:])
    # algebra of every returned edge
    ok = got["status"] == 1
    assert ok.mean() > 0.99
    dev_lin = []
    for i in np.nonzero(ok)[0][:64]:
        E, R, t = got["E"][i].reshape(3, 3), got["R"][i].reshape(3, 3), got["t"][i]
        assert abs(np.linalg.norm(E) - 1) < 1e-6 and abs(np.linalg.det(R) - 1) < 1e-9 and abs(t @ t - 1) < 1e-9
        Et = np.cross(np.eye(3), t) @ R
        Et /= np.linalg.norm(Et)
        # E is the essential matrix OF THE RETURNED POSE (round 6): [t]x R at unit norm, rank 2 whatever refit the fitted
        # model came from (after a LINEAR refit the model itself is only within a few 1e-3 of the manifold -- a caller
        # that builds F from E, matcher.h:216-217, must not get that)
        dev_lin.append(min(np.linalg.norm(Et - E), np.linalg.norm(Et + E)))
        assert dev_lin[-1] < 1e-9
        sv = np.linalg.svd(E, compute_uv=False)
        assert abs(sv[0] - sv[1]) < 1e-9 and sv[2] < 1e-9
        a0, a1 = int(b["offsets"][i]), int(b["offsets"][i + 1])
        assert got["n_inl"][i] == m[a0:a1].sum()
    errs = [S.rot_err_deg(got["R"][i].reshape(3, 3), b["R"][i]) if ok[i] else np.inf for i in range(512)]
    assert S.auc_at(errs) > 0.97
    print("||E - [t]x R|| at the default lo_linear_pct: median %.2e max %.2e" % (np.median(dev_lin), np.max(dev_lin)))
    # ... and with the Nister refit only (lo_linear_pct = 0)
    try:
        eng.set_params(lo_linear_pct=0)
        g0 = eng.edges_to_numpy(eng.estimate_pose_batch(db)[0])
        for i in np.nonzero(g0["status"] == 1)[0][:64]:
            E, R, t = g0["E"][i].reshape(3, 3), g0["R"][i].reshape(3, 3), g0["t"][i]
            Et = np.cross(np.eye(3), t) @ R
            Et /= np.linalg.norm(Et)
            assert min(np.linalg.norm(Et - E), np.linalg.norm(Et + E)) < 1e-9
    finally:
        eng.set_params(lo_linear_pct=35)
    # the oracle agrees on a bounded sample of the same workload
    exp, emasks = O.estimate_pose_batch(b["x1"][:a], b["y1"][:a], b["x2"][:a], b["y2"][:a], b["offsets"][:h + 1],
                                        7.5e-4, O.default_params(), 11, pair_id_base=20000)
    assert np.array_equal(m[:a], emasks)
    assert_edges_match(got[:h], exp)


def test_python_builder_surface():
    from pyposegraphbuilder import PoseGraphBuilder, findEssentialMatrix
    b = PoseGraphBuilder(20, 5000, 5, 100, 20, 50, 100, 0.8, 0.5, 0.4, "img", "ws", "sim.txt", "focals.txt", True, True, True)
    sizes = [300, 700, 40, 500]
    batch = S.make_batch(range(8100, 8104), sizes)
    pairs = []
    for i, n in enumerate(sizes):
        a, z = int(batch["offsets"][i]), int(batch["offsets"][i + 1])
        pairs.append(dict(src=i, dst=i + 1, similarity=0.9 - 0.1 * i if i != 3 else 0.2, threshold=7.5e-4,
                          correspondences=np.stack([batch[k][a:z] for k in ("x1", "y1", "x2", "y2")], 1)))
    g = b.run(pairs)
    assert set(g) == {(0, 1), (1, 2)}  # pair 2 has < kMinimumPointNumber matches, pair 3 is below the similarity threshold
    assert b.statistics["pairs_processed"] == 2 and b.statistics["edges_added"] == 2 and b.statistics["waves"] == 1
    for (s_, d_), e in g.items():
        assert S.rot_err_deg(e["R"], batch["R"][s_]) < 1.0 and 0.3 < e["score"] < 0.7
    ok, R, t, mask, n_inl = b.estimatePose(pairs[0]["correspondences"], 7.5e-4)
    assert ok and n_inl == mask.sum() and S.rot_err_deg(R, batch["R"][0]) < 1.0
    # historical binding shape: pixels + intrinsics in, (E, mask) out; bad shapes raise ValueError
    K = np.array([[1000.0, 0, 640], [0, 1000.0, 480], [0, 0, 1]])
    c = pairs[1]["correspondences"].astype(np.float64)
    E, m = findEssentialMatrix(c[:, :2] * 1000 + [640, 480], c[:, 2:] * 1000 + [640, 480], K, K, threshold=0.75)
    assert E is not None and m.dtype == bool and m.sum() > 300
    Egt = np.cross(np.eye(3), batch["t"][1]) @ batch["R"][1]
    Egt /= np.linalg.norm(Egt)
    assert min(np.linalg.norm(E - Egt), np.linalg.norm(E + Egt)) < 2e-2
    with pytest.raises(ValueError):
        findEssentialMatrix(np.zeros((5, 3)), np.zeros((5, 3)), K, K)


@pytest.mark.parametrize("kw", [dict(lo_iters=0), dict(lo_iters=1), dict(lo_iters=3), dict(round_size=16), dict(round_size=64),
                                dict(round_size=20), dict(confidence=0.999), dict(max_iters=100), dict(min_inliers=200),
                                dict(vote_all_rows=1), dict(guess_quirk=0), dict(fixed_budget=40, lo_iters=1),
                                dict(sampler=1), dict(sampler=1, max_iters=300, round_size=16)])
def test_parameter_variants_match_oracle(eng, kw):
    sizes = [300, 1000, 64, 2000, 150, 700, 450, 90]
    b = S.make_batch(range(12000, 12000 + len(sizes)), sizes, inlier_ratio=0.45)
    guesses = np.zeros((len(sizes), 12))
    has = np.zeros(len(sizes), np.uint8)
    for i in (1, 4):  # two pairs carry an (exact) pose guess
        guesses[i, :9], guesses[i, 9:] = b["R"][i].ravel(), b["t"][i]
        has[i] = 1
    base = dict(lo_iters=2, round_size=32, confidence=0.99, max_iters=1000, min_inliers=20, vote_all_rows=0,
                guess_quirk=1, fixed_budget=0, sampler=0)
    eng.set_params(**{**base, **kw})
    try:
        db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, guesses=guesses, has_guess=has,
                        seed=23, pair_id_base=12000)
        edges, masks = eng.estimate_pose_batch(db)
        got = eng.edges_to_numpy(edges)
    finally:
        eng.set_params(**base)
    exp, emasks = O.estimate_pose_batch(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4,
                                        O.default_params(**kw), 23, pair_id_base=12000, guesses=guesses, has_guess=has)
    assert np.array_equal(masks.cpu().numpy(), emasks)
    assert_edges_match(got, exp)


def test_progressive_sampling_on_ratio_sorted_rows(eng):
    """pgi_params.sampler = 1 (optional PROSAC): rows sorted by a quality score that correlates with being an
    inlier, as the matcher's ratio-sorted output does (feature_utils.h:184-186).  HIP == oracle bit for bit.  The schedule
    finds the model early; the stopping rule still counts inliers over all rows, so the gain shows where the budget is
    the limit: 35 % inliers need ~875 uniform hypotheses, the cap here is 160.  On unsorted rows nothing is lost."""
    rng = np.random.default_rng(77)
    P, N = 256, 1000
    b = S.make_batch(range(15000, 15000 + P), N, inlier_ratio=0.35)
    srt = {k: b[k].copy() for k in ("x1", "y1", "x2", "y2")}
    for i in range(P):
        a, z = i * N, (i + 1) * N
        quality = np.where(b["inlier"][a:z], rng.random(N) * 0.8, 0.2 + rng.random(N) * 0.8)  # lower = better match
        order = a + np.argsort(quality, kind="stable")
        for k in srt:
            srt[k][a:z] = b[k][order]
    stats = {}
    try:
        for tag, rows in (("sorted", srt), ("unsorted", b)):
            for sampler in (0, 1):
                eng.set_params(sampler=sampler, max_iters=160)
                db = eng.upload(rows["x1"], rows["y1"], rows["x2"], rows["y2"], b["offsets"], 7.5e-4, seed=5, pair_id_base=15000)
                edges, masks = eng.estimate_pose_batch(db)
                got = eng.edges_to_numpy(edges)
                exp, emasks = O.estimate_pose_batch(rows["x1"], rows["y1"], rows["x2"], rows["y2"], b["offsets"], 7.5e-4,
                                                    O.default_params(sampler=sampler, max_iters=160), 5, pair_id_base=15000)
                assert np.array_equal(masks.cpu().numpy(), emasks)
                assert_edges_match(got, exp)
                err = np.array([S.rot_err_deg(got["R"][i].reshape(3, 3), b["R"][i]) if got["status"][i] == 1 else 180.0 for i in range(P)])
                stats[(tag, sampler)] = (got["iters"].mean(), np.mean(err < 1.0))
    finally:
        eng.set_params(sampler=0, max_iters=1000)
    print("hypotheses / recall@1deg:", {k: (round(float(v[0]), 1), round(float(v[1]), 3)) for k, v in stats.items()})
    assert stats[("sorted", 1)][1] > stats[("sorted", 0)][1] + 0.1            # the point of the schedule
    assert abs(stats[("unsorted", 1)][1] - stats[("unsorted", 0)][1]) < 0.12   # without an ordering: neither gain nor harm


def test_understated_max_corr_is_reported_not_overrun(eng):
    b = S.make_batch([13000, 13001], [200, 900])
    db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, seed=1)
    db["max_corr"] = 256  # lie: the second pair has 900 rows
    edges, masks = eng.estimate_pose_batch(db)
    got = eng.edges_to_numpy(edges)
    assert got["status"][0] == 1 and got["status"][1] == -3
    assert masks.cpu().numpy()[200:].sum() == 0


def test_degenerate_inputs_match_oracle_and_terminate(eng):
    """Planar scene, pure rotation (E undefined), duplicated rows, NaN / huge coordinates: no hang, same result as the oracle."""
    r = np.random.default_rng(11)
    n = 400
    cases = []
    # planar scene (a known degenerate configuration for the five-point problem)
    d = S.make_pair(14000, n, inlier_ratio=0.6)
    Rg, tg = d["R"], d["t"]
    X = np.stack([r.uniform(-1.5, 1.5, n), r.uniform(-1.5, 1.5, n), np.full(n, 5.0)], 1)
    Y = X @ Rg.T + tg
    cases.append(np.concatenate([X[:, :2] / X[:, 2:], Y[:, :2] / Y[:, 2:]], 1))
    # pure rotation: t = 0
    Y = X * [1, 1, 1] @ Rg.T
    Xr = np.stack([r.uniform(-2, 2, n), r.uniform(-2, 2, n), r.uniform(3, 8, n)], 1)
    Yr = Xr @ Rg.T
    cases.append(np.concatenate([Xr[:, :2] / Xr[:, 2:], Yr[:, :2] / Yr[:, 2:]], 1))
    # every row identical
    cases.append(np.tile([[0.1, -0.2, 0.15, -0.18]], (n, 1)))
    # only 7 distinct rows, repeated
    base = np.stack([d[k][:7] for k in ("x1", "y1", "x2", "y2")], 1)
    cases.append(np.tile(base, (n // 7 + 1, 1))[:n])
    # NaN and huge values mixed into an otherwise good pair
    good = np.stack([d[k] for k in ("x1", "y1", "x2", "y2")], 1).astype(np.float64)
    bad = good.copy()
    bad[::17, 0] = np.nan
    bad[5::23, 3] = 1e30
    bad[7::29, 1] = -np.inf
    cases.append(bad)
    # all zeros
    cases.append(np.zeros((n, 4)))
    c = np.concatenate(cases).astype(np.float32)
    off = np.arange(len(cases) + 1) * n
    db = eng.upload(c[:, 0], c[:, 1], c[:, 2], c[:, 3], off, 7.5e-4, seed=3, pair_id_base=14000)
    edges, masks = eng.estimate_pose_batch(db)
    got = eng.edges_to_numpy(edges)
    exp, emasks = O.estimate_pose_batch(c[:, 0].copy(), c[:, 1].copy(), c[:, 2].copy(), c[:, 3].copy(), off, 7.5e-4,
                                        O.default_params(), 3, pair_id_base=14000)
    assert np.array_equal(masks.cpu().numpy(), emasks)
    assert list(got["status"]) == list(exp["status"]) and list(got["n_inl"]) == list(exp["n_inl"])
    assert list(got["iters"]) == list(exp["iters"])
    ok = exp["status"] == 1
    assert np.array_equal(got["E"][ok], exp["E"][ok])
    np.testing.assert_allclose(got["R"][ok], exp["R"][ok], atol=1e-9)
    # the contaminated pair still recovers the pose from its clean rows
    assert got["status"][4] == 1 and S.rot_err_deg(got["R"][4].reshape(3, 3), Rg) < 0.5


def test_size_bucketed_launches_match_oracle(eng):
    """>= 64 pairs spanning every occupancy class of launch_estimate (csrc/pgi_kernels.hip; all classes are 256-thread /
    four-wavefront workgroups): <= 1344 rows staged in LDS at four workgroups per CU; <= 2176 the hybrid class (1280 rows in
    LDS, the tail from HBM/L2, still four per CU); <= 3904 in LDS at two per CU; larger pairs read every row from HBM/L2.
    The classes run as persistent grids (resident workgroups pull pairs from the class's list), with PGI_K1_PERSISTENT=0 as
    one workgroup per list entry: the same bytes either way."""
    sizes = ([60, 300, 2300, 1500, 4000, 2176, 2177, 900] * 9)[:68] + [9100, 9024, 7000, 7900]
    ids = np.arange(15000, 15000 + len(sizes))
    b = S.make_batch(ids, sizes)
    db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, seed=4, pair_id_base=15000)
    edges, masks = eng.estimate_pose_batch(db)
    got = eng.edges_to_numpy(edges)
    exp, emasks = O.estimate_pose_batch(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4,
                                        O.default_params(), 4, pair_id_base=15000)
    assert np.array_equal(masks.cpu().numpy(), emasks)
    assert_edges_match(got, exp)
    # a second call reuses the bucket workspace and is bit-identical
    edges2, masks2 = eng.estimate_pose_batch(db)
    assert np.array_equal(edges.cpu().numpy(), edges2.cpu().numpy())
    # the one-workgroup-per-entry form of the class launches (a fresh context reads the switch) gives the same bytes
    import os
    from pyposegraphbuilder import Engine
    os.environ["PGI_K1_PERSISTENT"] = "0"
    try:
        eng2 = Engine()
    finally:
        del os.environ["PGI_K1_PERSISTENT"]
    try:
        db2 = eng2.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, seed=4, pair_id_base=15000)
        edges3, masks3 = eng2.estimate_pose_batch(db2)
        assert np.array_equal(edges.cpu().numpy(), edges3.cpu().numpy()) and np.array_equal(masks.cpu().numpy(), masks3.cpu().numpy())
    finally:
        eng2.close()


@pytest.mark.gpu
def test_host_buffer_batch_equals_device_batch():
    """pgi_estimate_pose_batch_host (chunked, two streams, copies overlapping kernels) == one launch on resident data."""
    from pyposegraphbuilder import Engine
    rng = np.random.default_rng(5150)
    P = 3000
    sizes = rng.choice([5, 40, 64, 300, 700, 1500, 2600, 4100], P)   # > 2.5 M rows: several chunks, ragged buckets
    b = S.make_batch(np.arange(40000, 40000 + P), sizes)
    guesses = np.zeros((P, 12))
    has = (rng.random(P) < 0.2).astype(np.uint8)
    for i in np.nonzero(has)[0]:
        guesses[i, :9], guesses[i, 9:] = b["R"][i].ravel(), b["t"][i]
    eng = Engine()
    try:
        db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, guesses=guesses, has_guess=has, seed=11,
                        pair_id_base=123)
        e, m = eng.estimate_pose_batch(db)
        ref, ref_m = eng.edges_to_numpy(e), m.cpu().numpy()
        got, got_m = eng.estimate_pose_batch_host(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, guesses=guesses,
                                                  has_guess=has, seed=11, pair_id_base=123)
        assert np.array_equal(got_m, ref_m)
        for k in ref.dtype.names:
            assert np.array_equal(got[k], ref[k]), k
        # without guesses, and a single tiny chunk
        got2, m2 = eng.estimate_pose_batch_host(b["x1"][:int(b["offsets"][7])], b["y1"][:int(b["offsets"][7])],
                                                b["x2"][:int(b["offsets"][7])], b["y2"][:int(b["offsets"][7])], b["offsets"][:8], 7.5e-4,
                                                seed=11, pair_id_base=123)
        db2 = eng.upload(b["x1"][:int(b["offsets"][7])], b["y1"][:int(b["offsets"][7])], b["x2"][:int(b["offsets"][7])],
                         b["y2"][:int(b["offsets"][7])], b["offsets"][:8], 7.5e-4, seed=11, pair_id_base=123)
        e2, mm2 = eng.estimate_pose_batch(db2)
        assert np.array_equal(eng.edges_to_numpy(e2)["E"], got2["E"]) and np.array_equal(mm2.cpu().numpy(), m2)
    finally:
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("direct", ["1", "0"])
def test_page_locked_host_batch_equals_device_batch(direct, monkeypatch):
    """Page-locked inputs and results: K1 works on them in place over PCIe (rows beyond the LDS part go through a device
    mirror); PGI_HOST_DIRECT=0 pipelines copies through HBM instead.  Both must equal one launch on resident data -- every size class (LDS,
    hybrid, two-workgroup LDS, rows from L2), with and without guesses, uniform small pairs (no mirror) and ragged ones."""
    from pyposegraphbuilder import Engine
    monkeypatch.setenv("PGI_HOST_DIRECT", direct)
    rng = np.random.default_rng(77)
    eng = Engine()
    try:
        for case, (P, choices) in enumerate([(2600, [5, 40, 64, 300, 700, 1300, 1500, 2100, 2600, 4100, 9000]), (900, [200, 640, 1000])]):
            sizes = rng.choice(choices, P)
            b = S.make_batch(np.arange(7000, 7000 + P), sizes)
            guesses = np.zeros((P, 12))
            has = (rng.random(P) < 0.2).astype(np.uint8)
            for i in np.nonzero(has)[0]:
                guesses[i, :9], guesses[i, 9:] = b["R"][i].ravel(), b["t"][i]
            for use_guess in ([True, False] if case == 0 else [False]):
                kw = dict(guesses=guesses, has_guess=has) if use_guess else {}
                db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, seed=3, pair_id_base=55, **kw)
                e, m = eng.estimate_pose_batch(db)
                ref, ref_m = eng.edges_to_numpy(e), m.cpu().numpy()
                xs = [np.array(b[k], np.float32) for k in ("x1", "y1", "x2", "y2")]
                out = (np.zeros(P, ref.dtype), np.zeros(len(ref_m), np.uint8))
                eng.pin(*xs, *out)
                try:
                    for rep in range(2):   # the second call reuses the mirror and the staging block
                        out[0][:] = 0
                        out[1][:] = 7
                        got, got_m = eng.estimate_pose_batch_host(*xs, b["offsets"], 7.5e-4, seed=3, pair_id_base=55, out=out, **kw)
                        assert np.array_equal(got_m, ref_m)
                        for k in ref.dtype.names:
                            assert np.array_equal(got[k], ref[k]), k
                    # page-locked inputs, pageable results: the copy pipeline
                    got, got_m = eng.estimate_pose_batch_host(*xs, b["offsets"], 7.5e-4, seed=3, pair_id_base=55, **kw)
                    assert np.array_equal(got_m, ref_m) and np.array_equal(got["E"], ref["E"])
                finally:
                    eng.unpin(*xs, *out)
    finally:
        eng.close()


def test_partially_page_locked_buffers_are_refused(eng):
    """A caller may have page-locked a shorter range than the batch needs (a pinned slice of a larger array).  The HIP
    runtime rejects copies that leave a registered range and a kernel working in place would fault past its end, so the
    call checks both ends of every range up front and fails loudly; fully registered and fully pageable buffers work."""
    from pyposegraphbuilder import _lib as L
    P = 700
    b = S.make_batch(np.arange(9100, 9100 + P), 600)
    db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, seed=5, pair_id_base=9)
    e, m = eng.estimate_pose_batch(db)
    ref, ref_m = eng.edges_to_numpy(e), m.cpu().numpy()
    xs = [np.array(b[k], np.float32) for k in ("x1", "y1", "x2", "y2")]
    out = (np.zeros(P, ref.dtype), np.zeros(len(ref_m), np.uint8))
    for part in ([xs[0][:len(xs[0]) // 2]], [out[1][:len(out[1]) // 3]], [xs[2][len(xs[2]) // 2:]]):   # a head, a head, a tail
        eng.pin(*part)
        try:
            with pytest.raises(L.PgiError, match="only in part"):
                eng.estimate_pose_batch_host(*xs, b["offsets"], 7.5e-4, seed=5, pair_id_base=9, out=out)
        finally:
            eng.unpin(*part)
    got, got_m = eng.estimate_pose_batch_host(*xs, b["offsets"], 7.5e-4, seed=5, pair_id_base=9, out=out)
    assert np.array_equal(got_m, ref_m) and np.array_equal(got["E"], ref["E"])


def test_rotation_guided_guess_mode_matches_oracle(eng):
    """guess_mode = 1 (BASELINE config 5): keep the guess's rotation, re-estimate the translation direction
    from 32 two-point hypotheses, local optimisation, accept at min_inliers, else the robust fit -- bit-identical to the
    oracle, and a chained pose with a good rotation but a meaningless translation now yields the right edge (the
    reference's own guess path, mode 0, accepts garbage there because of its un-squared inlier bound)."""
    sizes = [300, 1000, 64, 2000, 150, 2500, 700, 90]
    rhos = [0.5, 0.3, 0.6, 0.7, 0.2, 0.5, 0.1, 0.5]
    ids = np.arange(9100, 9100 + len(sizes))
    parts = [S.make_pair(int(i), n, inlier_ratio=r) for i, n, r in zip(ids, sizes, rhos)]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    cat = lambda k: np.concatenate([p[k] for p in parts])
    rng = np.random.default_rng(5)
    guesses = np.zeros((len(sizes), 12))
    has = np.ones(len(sizes), np.uint8)
    for i, p in enumerate(parts):
        Rg = S.rodrigues(rng.standard_normal(3), np.deg2rad(0.5)) @ p["R"]       # chained rotation: half a degree off
        if i == 2:
            Rg = S.rodrigues(rng.standard_normal(3), np.deg2rad(40.0)) @ p["R"]  # a wrong rotation: falls back to the robust fit
        guesses[i] = np.r_[Rg.ravel(), rng.standard_normal(3)]                    # translation: meaningless
    has[5] = 0                                                                    # one pair without a guess
    thr = 7.5e-4
    db = eng.upload(cat("x1"), cat("y1"), cat("x2"), cat("y2"), off, thr, guesses=guesses, has_guess=has, seed=21, pair_id_base=400)
    try:
        eng.set_params(guess_mode=1)
        edges, masks = eng.estimate_pose_batch(db)
        got, gm = eng.edges_to_numpy(edges), masks.cpu().numpy()
        exp, em = O.estimate_pose_batch(cat("x1"), cat("y1"), cat("x2"), cat("y2"), off, thr, O.default_params(guess_mode=1), 21,
                                        pair_id_base=400, guesses=guesses, has_guess=has)
        assert np.array_equal(gm, em)
        for f in ("E", "status", "n_inl", "score", "iters", "used_guess", "lo_runs", "cand", "votes"):
            assert np.array_equal(got[f], exp[f]), f
        np.testing.assert_allclose(got["R"], exp["R"], atol=1e-12)
        assert list(got["used_guess"]) == [1, 1, 0, 1, 1, 0, got["used_guess"][6], 1] and got["iters"][0] == 32
        for i in (0, 1, 3, 4, 7):
            assert S.rot_err_deg(got["R"][i].reshape(3, 3), parts[i]["R"]) < 2.0   # pair 4 has ~30 inliers of 150 rows
        # the reference's guess path on the same input accepts the chained pose's garbage
        eng.set_params(guess_mode=0)
        e0 = eng.edges_to_numpy(eng.estimate_pose_batch(db)[0])
        bad = [S.rot_err_deg(e0["R"][i].reshape(3, 3), parts[i]["R"]) for i in (0, 1, 3) if e0["status"][i] == 1]
        assert max(bad) > 5.0
    finally:
        eng.set_params(guess_mode=0)


def test_host_scoring_seam_from_twenty_threads(eng):
    """pgi_score_pose_f64_host -- the seam EssentialMatrixEvaluator::getInliers (graph_traversal.h:136-168) and
    InTraversalPoseTester::test (:194-233) sit behind inside A* -- called like the reference calls it: 20 threads, 1000 calls
    each, host pointers, no shared state.  Every answer equals the literal restatements pgo_ref_get_inliers /
    pgo_ref_pose_test (index lists; (true, kMin) at the early exit, (false, count) otherwise)."""
    import threading
    sizes = [50, 64, 257, 600, 1000, 2000, 5, 1]
    b = S.make_batch(range(8800, 8800 + len(sizes)), sizes)
    thr = 7.5e-4
    cases = []
    rng = np.random.default_rng(12)
    for p, n in enumerate(sizes):
        a, z = int(b["offsets"][p]), int(b["offsets"][p + 1])
        corr = np.stack([b["x1"][a:z], b["y1"][a:z], b["x2"][a:z], b["y2"][a:z]], 1).astype(np.float64)
        good = (b["R"][p], b["t"][p])
        ax = rng.standard_normal(3)
        bad = (S.rodrigues(ax / np.linalg.norm(ax), 0.4) @ b["R"][p], b["t"][p][::-1].copy())
        for R, t in (good, bad):
            E = O.ref_essential_from_pose(R, t)
            for kmin in (5, 20, 100000):
                ok, cnt = O.ref_pose_test(corr, R, t, 1.5 * thr, kmin)
                cases.append(("test", corr, E, (1.5 * thr) ** 2, kmin, (ok, cnt)))
            for tau in (1.5 * thr, (1.5 * thr) ** 2):          # the un-squared quirk bound (:164) and the squared one
                cases.append(("inliers", corr, E, tau, 0, O.ref_get_inliers(corr, E, tau)))
    errors = []

    def worker(tid):
        r = np.random.default_rng(100 + tid)
        try:
            for _ in range(1000):
                kind, corr, E, tau2, kmin, exp = cases[int(r.integers(len(cases)))]
                if kind == "test":
                    reached, cnt, _ = eng.score_pose_host(corr, E, tau2, early_exit_at=kmin, want_mask=False)
                    if (reached, cnt) != exp:
                        errors.append((tid, kind, len(corr), kmin, (reached, cnt), exp))
                else:
                    reached, cnt, mask = eng.score_pose_host(corr, E, tau2)
                    if reached or cnt != len(exp) or not np.array_equal(np.nonzero(mask)[0], exp):
                        errors.append((tid, kind, len(corr), cnt, len(exp)))
        except Exception as ex:  # noqa: BLE001
            errors.append((tid, repr(ex)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(20)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]
    # with a mask wanted the scan is complete even when an early-exit level is given
    kind, corr, E, tau2, _, exp = next(c for c in cases if c[0] == "inliers" and len(c[1]) == 2000)
    reached, cnt, mask = eng.score_pose_host(corr, E, tau2, early_exit_at=5)
    assert reached and cnt == 5 and np.array_equal(np.nonzero(mask)[0], exp)
    # empty input
    assert eng.score_pose_host(np.zeros((0, 4)), np.eye(3), 1.0, early_exit_at=5, want_mask=False)[:2] == (False, 0)


def test_config2_at_full_size_equals_the_oracle(eng):
    """BASELINE config 2 in the suite at its stated size -- all 10 000 pairs x 2 000 correspondences, bench.py's own ids and
    seed -- against the CPU oracle on every pair (OpenMP over the box's cores: about a second): identical masks, models,
    counts, iteration and refit numbers; R within 1e-4 rad, t-direction cosine within 1e-3 (north_star's bars)."""
    P, N, seed, thr = 10000, 2000, 0xB0BA, 7.5e-4
    b = S.make_batch(np.arange(P), N)
    db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], thr, seed=seed, pair_id_base=0)
    edges, masks = eng.estimate_pose_batch(db)
    got = eng.edges_to_numpy(edges)
    exp, emasks = O.estimate_pose_batch(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], thr, O.default_params(), seed,
                                        pair_id_base=0)
    assert np.array_equal(masks.cpu().numpy(), emasks)                     # identical inlier masks, 20 M rows
    assert np.array_equal(got["E"], exp["E"])
    for k in ("status", "n_inl", "score", "iters", "lo_runs", "votes", "cand", "used_guess"):
        assert np.array_equal(got[k], exp[k]), k
    ok = exp["status"] == 1
    assert ok.sum() >= 0.999 * P
    Rg, Re = got["R"][ok].reshape(-1, 3, 3), exp["R"][ok].reshape(-1, 3, 3)
    ang = np.arccos(np.clip((np.einsum("kij,kij->k", Rg, Re) - 1) / 2, -1, 1))
    assert ang.max() < R_TOL_RAD
    assert np.einsum("ki,ki->k", got["t"][ok], exp["t"][ok]).min() > 1 - T_COS_TOL
    errs = np.array([S.rot_err_deg(got["R"][i].reshape(3, 3), b["R"][i]) if ok[i] else np.inf for i in range(P)])
    assert S.auc_at(errs) > 0.985


def test_config1_single_pair_of_2000_correspondences(eng):
    """BASELINE config 1 ("single pair, 2k synthetic corrs", the examples/cpp_example.cpp path): exactly N = 2000 through the
    literal seam pgi_estimate_pose (host pointers, cv::Mat N x 4 CV_64F layout), with and without a pose guess, against
    the oracle's pgo_estimate_pose."""
    d = S.make_pair(424242, 2000)
    corr = np.stack([d["x1"], d["y1"], d["x2"], d["y2"]], 1).astype(np.float64)
    assert corr.shape == (2000, 4)
    thr, seed, pid = 7.5e-4, 31, 424242
    guess = np.concatenate([d["R"].ravel(), d["t"]])
    for g in (None, guess):
        ok, e, mask = eng.estimate_pose(corr, thr, guesses=None if g is None else g[None], seed=seed, pair_id=pid)
        oe, omask = O.estimate_pose(d["x1"], d["y1"], d["x2"], d["y2"], thr, g, O.default_params(), seed, pid)
        assert ok and e.status == oe.status == 1
        assert np.array_equal(mask, omask) and np.array_equal(np.array(e.E), np.array(oe.E))
        assert (e.n_inl, e.iters, e.lo_runs, e.used_guess) == (oe.n_inl, oe.iters, oe.lo_runs, oe.used_guess)
        assert e.used_guess == (0 if g is None else 1)
        assert rot_angle(np.array(e.R), np.array(oe.R)) < R_TOL_RAD and np.array(e.t) @ np.array(oe.t) > 1 - T_COS_TOL
        assert S.rot_err_deg(np.array(e.R).reshape(3, 3), d["R"]) < (0.1 if g is None else 2.0)


@pytest.mark.parametrize("nw", [1, 2, 4])
def test_every_wavefront_count_per_pair_gives_the_oracles_bits(nw, monkeypatch):
    """K1 is a template over the wavefronts per image pair (round 5: one / two / four; launch_estimate picks by batch size,
    PGI_K1_NW forces).  A fit's hypotheses, scores and merges are order-free, so every count must give the ORACLE's bits:
    ragged pairs across the LDS and hybrid classes of each count (320 / 640 / 1344 rows whole in LDS, the rest hybrid or --
    four wavefronts, huge pairs -- from HBM/L2), tiny and degenerate pairs, then the same rows with pose guesses in both
    guess modes (the rotation-guided path deals its 32 two-point hypotheses out over the wavefronts)."""
    from pyposegraphbuilder import Engine
    monkeypatch.setenv("PGI_K1_NW", str(nw))
    e = Engine()
    try:
        sizes = ([60, 300, 321, 640, 641, 900, 1344, 1345, 2300, 4, 5, 64] * 6)[:70] + [4100, 9000]
        rhos = ([0.5, 0.3, 0.7, 0.5, 0.15, 0.6] * 12)[:len(sizes)]
        ids = np.arange(23000, 23000 + len(sizes))
        parts = [S.make_pair(int(i), n, inlier_ratio=r) for i, n, r in zip(ids, sizes, rhos)]
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
        cat = lambda k: np.concatenate([p[k] for p in parts])
        x1, y1, x2, y2 = cat("x1"), cat("y1"), cat("x2"), cat("y2")
        db = e.upload(x1, y1, x2, y2, off, 7.5e-4, seed=9, pair_id_base=23000)
        edges, masks = e.estimate_pose_batch(db)
        exp, em = O.estimate_pose_batch(x1, y1, x2, y2, off, 7.5e-4, O.default_params(), 9, pair_id_base=23000)
        assert np.array_equal(masks.cpu().numpy(), em)
        assert_edges_match(e.edges_to_numpy(edges), exp)
        rng = np.random.default_rng(nw)
        guesses = np.zeros((len(sizes), 12))
        has = (rng.random(len(sizes)) < 0.8).astype(np.uint8)
        for i, p in enumerate(parts):
            Rg = S.rodrigues(rng.standard_normal(3), np.deg2rad(0.5 if i % 7 else 40.0)) @ p["R"]
            guesses[i] = np.r_[Rg.ravel(), p["t"] if i % 3 else rng.standard_normal(3)]
        dbg = e.upload(x1, y1, x2, y2, off, 7.5e-4, guesses=guesses, has_guess=has, seed=9, pair_id_base=23000)
        for mode in (0, 1):
            e.set_params(guess_mode=mode)
            eg, mg = e.estimate_pose_batch(dbg)
            expg, emg = O.estimate_pose_batch(x1, y1, x2, y2, off, 7.5e-4, O.default_params(guess_mode=mode), 9, pair_id_base=23000,
                                              guesses=guesses, has_guess=has)
            assert np.array_equal(mg.cpu().numpy(), emg), mode
            assert_edges_match(e.edges_to_numpy(eg), expg)
    finally:
        e.close()


@pytest.mark.parametrize("nw,env", [(1, {"PGI_HYBRID_ROWS": "0"}), (2, {"PGI_HYBRID_ROWS": "0", "PGI_LDS_MIN_WGS": "4"}),
                                    (1, {"PGI_HYBRID_ROWS": "0", "PGI_LDS_MIN_WGS": "4"})])
def test_rows_from_memory_below_four_wavefronts_per_pair(nw, env, monkeypatch):
    """with one or two wavefronts per pair the rows-from-memory variant of K1 does not exist; experiment
    settings that route a class to it (PGI_HYBRID_ROWS=0, PGI_LDS_MIN_WGS=4) used to launch NOTHING and return success, leaving
    the records of those pairs unwritten.  Every pair must now carry the oracle's bits (the hybrid variant with an empty
    LDS part serves them)."""
    from pyposegraphbuilder import Engine
    monkeypatch.setenv("PGI_K1_NW", str(nw))
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    e = Engine()
    try:
        sizes = ([60, 300, 321, 640, 641, 900, 1344, 1345, 2300, 64] * 8)[:78] + [4100, 9000]
        b = S.make_batch(np.arange(31000, 31000 + len(sizes)), sizes)
        db = e.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, seed=4, pair_id_base=31000)
        edges = torch_full_edges(e, len(sizes))
        ed, mk = e.estimate_pose_batch(db, edges)
        exp, em = O.estimate_pose_batch(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, O.default_params(), 4, pair_id_base=31000)
        got = e.edges_to_numpy(ed)
        assert not np.any(got["status"] == 77), "records left unwritten"
        assert np.array_equal(mk.cpu().numpy(), em)
        assert_edges_match(got, exp)
    finally:
        e.close()


def torch_full_edges(e, P):
    """an edge buffer pre-filled with a status no kernel writes (77): an unwritten record shows"""
    import torch
    from pyposegraphbuilder import _lib as L
    host = np.zeros(P, L.EDGE_DTYPE)
    host["status"] = 77
    return torch.from_numpy(host.view(np.uint8).reshape(P, -1).copy()).to(e.device)


@pytest.mark.parametrize("nw", [4, 2, 1])
def test_graph_cut_local_optimisation_matches_the_oracle(nw, monkeypatch):
    """pgi_params.lo_graph_cut (the "GC" of GC-RANSAC): the refit's rows are the minimum cut of the spatial-coherence energy over
    the 4-D grid neighbourhood (oracle/pgi_oracle.c: pgo_gc_labels; tests/test_graph_cut.py checks that sweep against a generic
    max-flow).  On the device one wavefront builds the chains (collisions inside a 64-row step resolved by ballots) and runs the
    two sweeps; integer energies, so every result must be the oracle's bit for bit: ragged pairs of every size class at each
    wavefront count, rows that share cells on purpose (one pair is a 6 x 6 lattice repeated: long chains, many collisions per
    step), both lambda values, then the rotation-guided guess mode."""
    from pyposegraphbuilder import Engine
    monkeypatch.setenv("PGI_K1_NW", str(nw))
    e = Engine()
    try:
        sizes = ([60, 300, 321, 640, 900, 1344, 1345, 2300, 5, 64, 129] * 4)[:40] + [4100]
        rhos = ([0.5, 0.3, 0.7, 0.4] * 12)[:len(sizes)]
        ids = np.arange(41000, 41000 + len(sizes))
        parts = [S.make_pair(int(i), n, inlier_ratio=r) for i, n, r in zip(ids, sizes, rhos)]
        # one pair whose rows pile up in a few cells: snap its coordinates to a coarse lattice (plus the usual noise)
        lat = parts[7]
        for kx in ("x1", "y1", "x2", "y2"):
            lat[kx] = (np.round(lat[kx] * 6) / 6 + (lat[kx] - np.round(lat[kx] * 6) / 6) * 0.05).astype(np.float32)
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
        cat = lambda k: np.concatenate([p[k] for p in parts])
        x1, y1, x2, y2 = cat("x1"), cat("y1"), cat("x2"), cat("y2")
        db = e.upload(x1, y1, x2, y2, off, 7.5e-4, seed=21, pair_id_base=41000)
        base_e, _ = e.estimate_pose_batch(db)
        base = e.edges_to_numpy(base_e).copy()
        for lam in (9, 40):
            e.set_params(lo_graph_cut=lam)
            edges, masks = e.estimate_pose_batch(db)
            exp, em = O.estimate_pose_batch(x1, y1, x2, y2, off, 7.5e-4, O.default_params(lo_graph_cut=lam), 21, pair_id_base=41000)
            got = e.edges_to_numpy(edges)
            assert np.array_equal(masks.cpu().numpy(), em), lam
            assert_edges_match(got, exp)
            assert not np.array_equal(got["E"], base["E"])          # the mode is not a no-op
        rng = np.random.default_rng(nw)
        guesses = np.zeros((len(sizes), 12))
        has = (rng.random(len(sizes)) < 0.8).astype(np.uint8)
        for i, p in enumerate(parts):
            Rg = S.rodrigues(rng.standard_normal(3), np.deg2rad(0.5 if i % 7 else 40.0)) @ p["R"]
            guesses[i] = np.r_[Rg.ravel(), p["t"] if i % 3 else rng.standard_normal(3)]
        dbg = e.upload(x1, y1, x2, y2, off, 7.5e-4, guesses=guesses, has_guess=has, seed=21, pair_id_base=41000)
        for mode in (0, 1):
            e.set_params(guess_mode=mode, lo_graph_cut=9)
            eg, mg = e.estimate_pose_batch(dbg)
            expg, emg = O.estimate_pose_batch(x1, y1, x2, y2, off, 7.5e-4, O.default_params(guess_mode=mode, lo_graph_cut=9), 21,
                                              pair_id_base=41000, guesses=guesses, has_guess=has)
            assert np.array_equal(mg.cpu().numpy(), emg), mode
            assert_edges_match(e.edges_to_numpy(eg), expg)
        # the single-pair seam (pgi_estimate_pose: its own scratch inside the slot buffer)
        e.set_params(guess_mode=0, lo_graph_cut=9)
        a0, a1 = int(off[3]), int(off[4])
        corr = np.stack([x1[a0:a1], y1[a0:a1], x2[a0:a1], y2[a0:a1]], 1).astype(np.float64)
        ok, edge, m1 = e.estimate_pose(corr, 7.5e-4, seed=21, pair_id=41003)
        one, om = O.estimate_pose_batch(x1[a0:a1], y1[a0:a1], x2[a0:a1], y2[a0:a1], np.array([0, a1 - a0], np.uint64), 7.5e-4,
                                        O.default_params(lo_graph_cut=9), 21, pair_id_base=41003)
        assert np.array_equal(m1, om) and np.array_equal(np.array(edge.E), one["E"][0])
    finally:
        e.close()


def test_graph_cut_through_the_host_pointer_entries():
    """lo_graph_cut with host buffers in and out: pgi_estimate_pose_batch_host cuts the batch into chunks and hands every chunk's row
    count to the launcher (the two scratch words per row follow the chunk's size-bucket lists); page-locked buffers are NOT worked
    on in place in this mode (the labelling's second pass over the rows would fetch them over PCIe again) but copied like pageable
    ones.  Both must equal the resident launch, which equals the oracle."""
    from pyposegraphbuilder import Engine
    rng = np.random.default_rng(5)
    eng = Engine(lo_graph_cut=9)
    try:
        P = 2200
        sizes = rng.choice([5, 40, 64, 300, 700, 1300, 1500, 2100, 2600, 4100], P)
        b = S.make_batch(np.arange(61000, 61000 + P), sizes)
        db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, seed=3, pair_id_base=61000)
        e, m = eng.estimate_pose_batch(db)
        ref, ref_m = eng.edges_to_numpy(e), m.cpu().numpy()
        exp, em = O.estimate_pose_batch(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, O.default_params(lo_graph_cut=9), 3,
                                        pair_id_base=61000)
        assert np.array_equal(ref_m, em)
        assert_edges_match(ref, exp)
        xs = [np.array(b[k], np.float32) for k in ("x1", "y1", "x2", "y2")]
        got, got_m = eng.estimate_pose_batch_host(*xs, b["offsets"], 7.5e-4, seed=3, pair_id_base=61000)      # pageable
        assert np.array_equal(got_m, ref_m)
        for k in ref.dtype.names:
            assert np.array_equal(got[k], ref[k]), k
        out = (np.zeros(P, ref.dtype), np.zeros(len(ref_m), np.uint8))
        eng.pin(*xs, *out)
        try:
            got, got_m = eng.estimate_pose_batch_host(*xs, b["offsets"], 7.5e-4, seed=3, pair_id_base=61000, out=out)  # page-locked
            assert np.array_equal(got_m, ref_m)
            for k in ref.dtype.names:
                assert np.array_equal(got[k], ref[k]), k
        finally:
            eng.unpin(*xs, *out)
    finally:
        eng.close()


def test_graph_cut_single_pair_seam_from_many_threads():
    """pgi_estimate_pose with lo_graph_cut from 12 threads at once: every call leases its own slot, and the labelling's two scratch
    words per row live INSIDE the slot's device block -- nothing shared between concurrent callers.  Every answer equals the oracle."""
    import threading
    from pyposegraphbuilder import Engine
    sizes = [257, 600, 64, 1500, 333, 1024, 90, 2100, 700, 129, 1344, 45]
    b = S.make_batch(range(71000, 71000 + len(sizes)), sizes, inlier_ratio=0.55)
    exp, em = O.estimate_pose_batch(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, O.default_params(lo_graph_cut=9), 13,
                                    pair_id_base=71000)
    off = b["offsets"].astype(np.int64)
    eng = Engine(lo_graph_cut=9)
    errors = []

    def worker(t):
        try:
            for rep in range(4):
                p = (t + rep * 5) % len(sizes)
                a, z = int(off[p]), int(off[p + 1])
                corr = np.stack([b["x1"][a:z], b["y1"][a:z], b["x2"][a:z], b["y2"][a:z]], 1).astype(np.float64)
                ok, edge, m = eng.estimate_pose(corr, 7.5e-4, seed=13, pair_id=71000 + p)
                assert np.array_equal(m, em[a:z]) and np.array_equal(np.array(edge.E), exp["E"][p]) and int(edge.iters) == int(exp["iters"][p]), p
        except Exception as ex:  # noqa: BLE001
            errors.append((t, repr(ex)))
    try:
        ths = [threading.Thread(target=worker, args=(t,)) for t in range(12)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
    finally:
        eng.close()
    assert not errors, errors[:3]
