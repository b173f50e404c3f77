"""The reference's LITERAL candidate rule (pose_utils.h:144-169 decomposeEssentialMatrix, :172-252 getPoseFromEssentialMatrix,
:491-506 linearTriangulation) inside the suite.

CPU part: the restatements `pgo_ref_linear_triangulation` / `pgo_ref_decompose_essential` / `pgo_ref_pose_from_essential`
against INDEPENDENT numpy arithmetic and against the committed fixture (tests/golden/golden_v3_candidates.npz).
GPU part: `pgi_decompose_batch` (the product's depth-sign vote, default population and `vote_all_rows = 1`) against the
literal rule on 3 x 2000 seeded pairs: same rotation on every pair; same translation (null vector oriented w >= 0) on every
pair at inlier ratio >= 0.5; at 0.3 the agreement rates of the committed table (profiles/r02_candidate_agreement.md)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from pyposegraphbuilder import synthetic as S
from golden import make_golden_candidates as MG

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v3_candidates.npz"))
rng = np.random.default_rng(20261002)

# profiles/r02_candidate_agreement.md, inlier ratio 0.3 (32 211 pairs): product(vote_all_rows=1) == literal(w>=0) and
# product(inlier rows) t == literal(w>=0) t
TABLE_RHO03_ALL_ROWS = 0.966130
TABLE_RHO03_INLIER_ROWS_T = 0.906150


def skew(t):
    return np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])


def rand_rot(max_deg=180.0):
    ax = rng.standard_normal(3)
    return S.rodrigues(ax / np.linalg.norm(ax), np.radians(rng.uniform(0, max_deg)))


def dlt_matrix(P1, P2, pt):
    return np.stack([pt[0] * P1[2] - P1[0], pt[1] * P1[2] - P1[1], pt[2] * P2[2] - P2[0], pt[3] * P2[2] - P2[1]])


def test_ref_linear_triangulation_is_the_svd_null_vector():
    """pose_utils.h:491-506: X = last right singular vector of the 4x4 DLT matrix, sign not normalised.
    1000 random (P2, point) cases, a fifth of them with a near-degenerate baseline (|t| down to 1e-6)."""
    P1 = np.hstack([np.eye(3), np.zeros((3, 1))])
    worst = 0.0
    for k in range(1000):
        R = rand_rot(40.0)
        t = rng.standard_normal(3)
        t *= (10.0 ** rng.uniform(-6, -2) if k % 5 == 0 else 1.0) / np.linalg.norm(t)
        P2 = np.hstack([R, t[:, None]])
        Xw = np.array([*(rng.uniform(-0.45, 0.45, 2) * 4.0), rng.uniform(2, 8)])
        x2 = R @ Xw + t
        pt = np.array([Xw[0] / Xw[2], Xw[1] / Xw[2], x2[0] / x2[2], x2[1] / x2[2]]) + rng.normal(0, 2.5e-4, 4)
        X = O.ref_linear_triangulation(P1, P2, pt)
        D = dlt_matrix(P1, P2, pt)
        _, sv, Vt = np.linalg.svd(D)
        ref = Vt[3]
        assert abs(np.linalg.norm(X) - 1.0) < 1e-12
        # a null vector is defined up to sign; its conditioning is the gap to the next singular value
        gap = (sv[2] - sv[3]) / sv[0]
        err = min(np.abs(X - ref).max(), np.abs(X + ref).max())
        assert err < 1e-12 / max(gap, 1e-9), (k, err, gap)
        # and it minimises |D X| like numpy's does
        assert np.linalg.norm(D @ X) <= sv[3] * (1 + 1e-9) + 1e-15
        if gap > 1e-3:
            worst = max(worst, err)
    assert worst < 1e-12


def test_ref_decompose_essential_matches_numpy():
    """pose_utils.h:144-169: R1 = U d V^T, R2 = U d^T V^T after the det fixes, t = U.col(2).  An essential matrix has
    a double singular value, so U, V are defined up to a joint rotation of their first two columns (which leaves R1,
    R2 unchanged) and a joint flip (which swaps R1 <-> R2 and negates t): compare as the set {R1, R2} and t up to sign."""
    d = np.array([[0.0, 1, 0], [-1, 0, 0], [0, 0, 1]])
    for k in range(200):
        R, t = rand_rot(), rng.standard_normal(3)
        E = skew(t / np.linalg.norm(t)) @ R
        if k % 2:
            E = E + rng.normal(0, 1e-3, (3, 3))  # a noisy estimate: no longer exactly essential
        E *= rng.choice([-1.0, 1.0]) * 10.0 ** rng.uniform(-2, 2)
        R1, R2, tt = O.ref_decompose_essential(E)
        U, _, Vt = np.linalg.svd(E)
        if np.linalg.det(U) < 0:
            U[:, 2] = -U[:, 2]
        if np.linalg.det(Vt) < 0:
            Vt[2] = -Vt[2]
        N1, N2, nt = U @ d @ Vt, U @ d.T @ Vt, U[:, 2]
        same = max(np.abs(R1 - N1).max(), np.abs(R2 - N2).max(), np.abs(tt - nt).max())
        swapped = max(np.abs(R1 - N2).max(), np.abs(R2 - N1).max(), np.abs(tt + nt).max())
        assert min(same, swapped) < 1e-9, (k, same, swapped)
        for Rc in (R1, R2):
            assert np.abs(Rc @ Rc.T - np.eye(3)).max() < 1e-12 and abs(np.linalg.det(Rc) - 1) < 1e-12
        assert abs(np.linalg.norm(tt) - 1) < 1e-12
        if k % 2 == 0:  # exact essential matrix: the true rotation is one of the two, the true direction is +-t
            assert min(np.abs(R1 - R).max(), np.abs(R2 - R).max()) < 1e-9
            assert abs(abs(tt @ t) / np.linalg.norm(t) - 1) < 1e-12


def numpy_literal_rule(E, corr, orient):
    """pose_utils.h:172-252 with numpy's SVDs: per row and candidate a DLT point, raw-z tests, squared reprojection error,
    strict-< arg-min per row, one vote per row, first maximum wins.  orient: +1 null vector scaled to w >= 0, -1 w <= 0."""
    R1, R2, t = O.ref_decompose_essential(E)
    P1 = np.hstack([np.eye(3), np.zeros((3, 1))])
    cands = [(R1, t), (R1, -t), (R2, t), (R2, -t)]
    best_err = np.full(len(corr), np.inf)
    best_c = np.full(len(corr), 5)
    for c, (R, tt) in enumerate(cands):
        P2 = np.hstack([R, tt[:, None]])
        for p, pt in enumerate(corr):
            X = np.linalg.svd(dlt_matrix(P1, P2, pt))[2][3]
            X = X * (orient if X[3] >= 0 else -orient)
            p1, p2 = P1 @ X, P2 @ X
            if p1[2] < 0 or p2[2] < 0:
                continue
            err = ((p1[:2] / p1[2] - pt[:2]) ** 2).sum() + ((p2[:2] / p2[2] - pt[2:]) ** 2).sum()
            if err < best_err[p]:
                best_err[p], best_c[p] = err, c
    votes = np.array([(best_c == c).sum() for c in range(4)])
    return votes, int(np.argmax(votes))


def test_ref_pose_from_essential_matches_a_numpy_restatement():
    """the whole literal rule, oracle C against numpy SVDs, on noisy pairs with outliers"""
    for k, rho in enumerate((0.3, 0.6, 0.9, 0.5)):
        d = S.make_pair(7000 + k, 90, inlier_ratio=rho)
        corr = np.stack([d["x1"], d["y1"], d["x2"], d["y2"]], 1).astype(np.float64)
        E = skew(d["t"]) @ d["R"] * (-1) ** k
        R, t, votes, cand = O.ref_pose_from_essential(E, corr)
        for mode, orient in ((1, +1), (2, -1)):
            nv, nc = numpy_literal_rule(E, corr, orient)
            assert list(votes[mode]) == list(nv) and cand[mode] == nc
        # with w >= 0 the raw-z test is the cheirality test: the true pose wins; with w <= 0 its mirror image does
        assert np.abs(R[1] - d["R"]).max() < 1e-9 and np.abs(t[1] - d["t"]).max() < 1e-9
        assert np.abs(t[2] + d["t"]).max() < 1e-9
        # the raw convention's vote totals are a row-wise mixture of the two
        assert votes[0].sum() <= len(corr)


@pytest.mark.parametrize("k", [0, 1, 2])
def test_literal_rule_fixture(k):
    """the committed outputs of the literal rule (golden_v3_candidates.npz) are what the oracle computes today"""
    ids, sizes = G["fix%d_ids" % k], G["fix%d_sizes" % k]
    b = S.make_batch(ids, sizes, inlier_ratio=float(G["ratios"][k]))
    for i in range(0, len(ids), 4):
        if G["fix%d_status" % k][i] != 1:
            continue
        a, z = int(b["offsets"][i]), int(b["offsets"][i + 1])
        corr = np.stack([b["x1"][a:z], b["y1"][a:z], b["x2"][a:z], b["y2"][a:z]], 1).astype(np.float64)
        R, t, votes, cand = O.ref_pose_from_essential(G["fix%d_E" % k][i], corr)
        assert np.array_equal(cand, G["fix%d_cand" % k][i]) and np.array_equal(votes, G["fix%d_votes" % k][i])
        np.testing.assert_allclose(R.reshape(3, 9), G["fix%d_R" % k][i], atol=1e-12)
        np.testing.assert_allclose(t, G["fix%d_t" % k][i], atol=1e-12)


def test_fixture_rates_agree_with_the_committed_table():
    n, ok, r_inl, t_inl, r_all, t_all = (int(v) for v in G["live0_counts"])
    assert r_inl == ok and r_all == ok
    assert abs(t_all / ok - TABLE_RHO03_ALL_ROWS) < 0.01
    assert abs(t_inl / ok - TABLE_RHO03_INLIER_ROWS_T) < 0.01
    for k in (1, 2):
        assert len(set(int(v) for v in G["live%d_counts" % k])) == 1  # every pair OK, every pair agrees in R and t


# ---------------------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def eng():
    from pyposegraphbuilder import Engine
    e = Engine()
    yield e
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k", [0, 1, 2])
def test_gpu_decompose_batch_against_the_literal_rule(eng, k):
    """pgi_decompose_batch (HIP) vs pose_utils.h:172-252 restated literally, 2000 pairs per inlier ratio, fixed seeds."""
    rho = float(G["ratios"][k])
    ids = MG.pair_ids(rho, MG.N_LIVE)
    sizes = S.ragged_sizes(ids, 120, 420)
    b = S.make_batch(ids, sizes, inlier_ratio=rho)
    db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], float(G["thr"]), seed=int(G["seed"]),
                    pair_id_base=int(ids[0]))
    edges, masks = eng.estimate_pose_batch(db)
    est = eng.edges_to_numpy(edges)
    ok = np.nonzero(est["status"] == 1)[0]
    # the estimator's own epilogue and the stand-alone K3 launch are the same rule: the record's E is the essential matrix of
    # the returned pose ([t]x R, round 6), so decomposing it gives that pose back (to rounding: the epilogue decomposed the
    # fitted f32 model, K3 here the rebuilt f64 matrix) with the same candidate chosen by the same inlier votes
    dec = eng.edges_to_numpy(eng.decompose_batch(db, est["E"], masks))
    assert np.abs(dec["R"][ok] - est["R"][ok]).max() < 1e-9 and np.abs(dec["t"][ok] - est["t"][ok]).max() < 1e-9
    eng.set_params(vote_all_rows=1)
    try:
        dec_all = eng.edges_to_numpy(eng.decompose_batch(db, est["E"], masks))
    finally:
        eng.set_params(vote_all_rows=0)
    # the fixture's E came from the CPU oracle: the HIP estimator must have returned the same matrices
    f = slice(0, MG.N_FIXTURE)
    assert np.array_equal(est["status"][f], G["fix%d_status" % k]) and np.array_equal(est["E"][f], G["fix%d_E" % k])

    same_R = same_t = same_R_all = same_t_all = 0
    for i in ok:
        a, z = int(b["offsets"][i]), int(b["offsets"][i + 1])
        corr = np.stack([b["x1"][a:z], b["y1"][a:z], b["x2"][a:z], b["y2"][a:z]], 1).astype(np.float64)
        if i < MG.N_FIXTURE:
            lit_R, lit_t = G["fix%d_R" % k][i, 1], G["fix%d_t" % k][i, 1]
        else:
            R3, t3, _, _ = O.ref_pose_from_essential(est["E"][i], corr)
            lit_R, lit_t = R3[1].ravel(), t3[1]
        same_R += np.abs(dec["R"][i] - lit_R).max() < 1e-6
        same_t += np.abs(dec["t"][i] - lit_t).max() < 1e-6
        same_R_all += np.abs(dec_all["R"][i] - lit_R).max() < 1e-6
        same_t_all += np.abs(dec_all["t"][i] - lit_t).max() < 1e-6
    n = len(ok)
    assert [MG.N_LIVE, n, same_R, same_t, same_R_all, same_t_all] == [int(v) for v in G["live%d_counts" % k]]
    assert same_R == n and same_R_all == n               # the literal rule's rotation on 100 % of the pairs
    if rho >= 0.5:
        assert same_t == n and same_t_all == n           # and its translation (w >= 0)
    else:
        assert abs(same_t_all / n - TABLE_RHO03_ALL_ROWS) < 0.01
        assert abs(same_t / n - TABLE_RHO03_INLIER_ROWS_T) < 0.01


@pytest.mark.gpu
def test_pose_from_essential_host_seam_from_twenty_threads(eng):
    """pgi_pose_from_essential_host -- the seam pose::getPoseFromEssentialMatrix (pose_utils.h:172-252) sits behind -- called
    like the reference would call it: host pointers (E, the N x 4 CV_64F matrix), 20 threads x 200 calls on one context.
    Every answer equals pgi_decompose_batch on the same rows (all rows voting, and the inlier rows through h_mask) bit for
    bit, and the literal restatement pgo_ref_pose_from_essential (null vector oriented w >= 0): the same rotation on every
    pair, the same translation at inlier ratio >= 0.5."""
    import threading
    sizes = [120, 257, 600, 64, 2000, 333, 5, 1024]
    b = S.make_batch(range(9100, 9100 + len(sizes)), sizes, inlier_ratio=0.6)
    db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, seed=3, pair_id_base=9100)
    edges, masks = eng.estimate_pose_batch(db)
    est = eng.edges_to_numpy(edges)
    hm = masks.cpu().numpy()
    dec_in = eng.edges_to_numpy(eng.decompose_batch(db, est["E"], masks))
    dec_all = eng.edges_to_numpy(eng.decompose_batch(db, est["E"], None))
    cases = []
    for p in range(len(sizes)):
        a, z = int(b["offsets"][p]), int(b["offsets"][p + 1])
        corr = np.stack([b["x1"][a:z], b["y1"][a:z], b["x2"][a:z], b["y2"][a:z]], 1).astype(np.float64)
        lit = O.ref_pose_from_essential(est["E"][p], corr) if est["status"][p] == 1 else None
        cases.append((corr, est["E"][p], hm[a:z].copy(), dec_all[p], dec_in[p], lit, est["status"][p] == 1))
    errors = []

    def worker(tid):
        r = np.random.default_rng(500 + tid)
        try:
            for _ in range(200):
                corr, E, m, d_all, d_in, lit, ok = cases[int(r.integers(len(cases)))]
                if r.random() < 0.5:
                    R, t, votes, cand = eng.pose_from_essential_host(E, corr)
                    exp = d_all
                else:
                    R, t, votes, cand = eng.pose_from_essential_host(E, corr, mask=m)
                    exp = d_in
                if not (np.array_equal(R.ravel(), exp["R"], equal_nan=True) and np.array_equal(t, exp["t"], equal_nan=True) and votes == exp["votes"] and cand == exp["cand"]):
                    errors.append((tid, len(corr), votes, int(exp["votes"]), cand, int(exp["cand"])))
                if ok and lit is not None and exp is d_all:
                    if np.abs(R.ravel() - lit[0][1].ravel()).max() > 1e-6 or np.abs(t - lit[1][1]).max() > 1e-6:
                        errors.append((tid, "literal rule differs", len(corr)))
        except Exception as ex:  # noqa: BLE001
            errors.append((tid, repr(ex)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(20)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]
    # empty matrix: no votes, candidate 0 of the decomposition
    R, t, votes, cand = eng.pose_from_essential_host(est["E"][0], np.zeros((0, 4)))
    assert votes == 0 and cand == 0 and abs(np.linalg.det(R) - 1) < 1e-9 and abs(np.linalg.norm(t) - 1) < 1e-9
