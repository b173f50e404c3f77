"""Guided matching with a known pose (matcher.h:199-405, pose_graph_builder.h:715-783).

CPU: the oracle's fundamental matrix against numpy and its candidate gate against an independent restatement.
GPU: pgi_guided_match_batch through the C ABI == the oracle, index for index and ratio bit for bit."""
import numpy as np
import pytest

import oracle_lib as O
from pyposegraphbuilder import synthetic as S


def scene(seed, n_points=3000, n_clutter=3000, desc_noise=0.012):
    rng = np.random.default_rng(seed)
    views, poses, cam = S.make_feature_views(rng, n_views=3, n_points=n_points, n_clutter=n_clutter, desc_noise=desc_noise)
    return views, poses, cam


def rel_pose(poses, s, d):
    R = poses[d][0] @ poses[s][0].T
    return R, poses[d][1] - R @ poses[s][1]


def oracle_matches(views, poses, cam, s, d, max_n):
    R, t = rel_pose(poses, s, d)
    E = np.zeros(9)
    O.lib().pgo_ref_essential_from_pose(O._p(O.f64(R).ravel()), O._p(O.f64(t)), O._p(E))
    k = [cam[0], cam[0], cam[1] / 2.0, cam[2] / 2.0]
    F = O.fundamental_from_essential(E, k, k)
    oi, oj, orr = O.guided_match(F, views[s]["xy"], views[d]["xy"], views[s]["desc"], views[d]["desc"])
    if max_n and len(oi) > max_n:                       # pose_graph_builder.h:759-772: smallest (ratio, position) first
        order = np.lexsort((np.arange(len(oi)), orr))[:max_n]
        oi, oj, orr = oi[order], oj[order], orr[order]
    return oi, oj, orr, F


def test_oracle_fundamental_and_gate():
    views, poses, cam = scene(31, 1500, 1500)
    oi, oj, orr, F = oracle_matches(views, poses, cam, 0, 1, 0)
    k = np.array([[cam[0], 0, cam[1] / 2], [0, cam[0], cam[2] / 2], [0, 0, 1.0]])
    R, t = rel_pose(poses, 0, 1)
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    assert np.allclose(F.reshape(3, 3), np.linalg.inv(k).T @ (tx @ R) @ np.linalg.inv(k), rtol=1e-12, atol=1e-18)
    assert len(oi) > 100
    # every match obeys the epipolar gate (independent float64 evaluation) and the planted correspondences dominate
    x1 = np.c_[views[0]["xy"][oi].astype(float), np.ones(len(oi))]
    x2 = np.c_[views[1]["xy"][oj].astype(float), np.ones(len(oi))]
    Fm = F.reshape(3, 3)
    l2, l1 = x1 @ Fm.T, x2 @ Fm
    r = np.einsum("ij,ij->i", x2, l2)
    d = r * r * ((l1[:, 0] ** 2 + l1[:, 1] ** 2) + (l2[:, 0] ** 2 + l2[:, 1] ** 2)) / ((l1[:, 0] ** 2 + l1[:, 1] ** 2) * (l2[:, 0] ** 2 + l2[:, 1] ** 2))
    assert np.all(d < 0.75 ** 2 * (1 + 1e-9))
    pid0, pid1 = views[0]["point_id"][oi], views[1]["point_id"][oj]
    assert (pid0 == pid1).mean() > 0.98


@pytest.fixture(scope="module")
def eng():
    from pyposegraphbuilder import Engine
    e = Engine()
    yield e
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("max_n", [0, 100])
def test_gpu_guided_match_bit_exact(eng, max_n):
    views, poses, cam = scene(32)
    feats = [eng.upload_features(v["xy"], v["desc"], *cam) for v in views]
    pairs = [(0, 1), (1, 0), (0, 2), (2, 1)]
    rt = np.array([np.r_[rel_pose(poses, s, d)[0].ravel(), rel_pose(poses, s, d)[1]] for s, d in pairs])
    got = eng.guided_match_batch(feats, pairs, rt, max_n=max_n, n_bins=0)
    for (s, d), (gi, gj, gr) in zip(pairs, got):
        oi, oj, orr, _ = oracle_matches(views, poses, cam, s, d, max_n)
        assert len(oi) > (50 if max_n else 300)
        assert np.array_equal(gi, oi) and np.array_equal(gj, oj) and np.array_equal(gr, orr), (s, d)


@pytest.mark.gpu
def test_gpu_guided_match_degenerate_inputs(eng):
    """Empty images, a zero pose (F = 0: every distance is NaN, every destination point becomes a candidate and the
    per-thread queues overflow repeatedly), and a pure-rotation pose (E = 0)."""
    views, poses, cam = scene(33, 300, 300)
    feats = [eng.upload_features(v["xy"], v["desc"], *cam) for v in views]
    empty = eng.upload_features(np.zeros((0, 2), np.float32), np.zeros((0, 128), np.float32), *cam)
    feats.append(empty)
    pairs = [(0, 3), (3, 0), (0, 1), (1, 2)]
    rt = np.zeros((4, 12))
    rt[3, :9] = np.eye(3).ravel()
    got = eng.guided_match_batch(feats, pairs, rt, max_n=0, n_bins=0)
    assert len(got[0][0]) == 0 and len(got[1][0]) == 0
    for p in (2, 3):
        s, d = pairs[p]
        R, t = rt[p, :9].reshape(3, 3), rt[p, 9:]
        E = np.zeros(9)
        O.lib().pgo_ref_essential_from_pose(O._p(O.f64(R).ravel()), O._p(O.f64(t)), O._p(E))
        k = [cam[0], cam[0], cam[1] / 2.0, cam[2] / 2.0]
        oi, oj, orr = O.guided_match(O.fundamental_from_essential(E, k, k), views[s]["xy"], views[d]["xy"],
                                     views[s]["desc"], views[d]["desc"])
        assert np.array_equal(got[p][0], oi) and np.array_equal(got[p][1], oj) and np.array_equal(got[p][2], orr)


def oracle_matches_binned(views, poses, cam, s, d, max_n):
    R, t = rel_pose(poses, s, d)
    E = np.zeros(9)
    O.lib().pgo_ref_essential_from_pose(O._p(O.f64(R).ravel()), O._p(O.f64(t)), O._p(E))
    k = [cam[0], cam[0], cam[1] / 2.0, cam[2] / 2.0]
    F = O.fundamental_from_essential(E, k, k)
    size = (int(cam[1]), int(cam[2]))
    oi, oj, orr, frag = O.ref_guided_match_binned(F, views[s]["xy"], views[d]["xy"], views[s]["desc"], views[d]["desc"], size, size)
    return oi, oj, orr, frag


def test_oracle_binned_is_a_restriction_of_the_exhaustive_loop():
    """CPU: the literal epipolar hashing (matcher.h:218-331) only REMOVES candidates from the exhaustive loop: every
    binned match whose source point met all of its gate-passing candidates is the exhaustive match; a percent or so of
    the matches differ (a gate-passing destination keypoint fell into a neighbouring bin)."""
    views, poses, cam = scene(32)
    tot = same = 0
    for s, d in [(0, 1), (1, 0), (0, 2)]:
        oi, oj, orr, _ = oracle_matches(views, poses, cam, s, d, 0)
        bi, bj, br, frag = oracle_matches_binned(views, poses, cam, s, d, 0)
        ex = dict(zip(oi.tolist(), zip(oj.tolist(), orr.tolist())))
        bn = dict(zip(bi.tolist(), zip(bj.tolist(), br.tolist())))
        tot += len(ex)
        same += sum(1 for k in ex if bn.get(k) == ex[k])
        assert len(bn) > 0.95 * len(ex) and frag.sum() == 0
    assert 0.95 < same / tot < 1.0   # close, but NOT identical: hence the binned mode on the device


@pytest.mark.gpu
@pytest.mark.parametrize("max_n", [0, 100])
def test_gpu_guided_match_binned_equals_literal_restatement(eng, max_n):
    """pgi_guided_match_batch(n_bins = 45) == pgo_ref_guided_match_binned (the reference's epipolar hashing, restated
    literally with libm atan2 / round) for every source keypoint outside the don't-care band: a source point is
    'fragile' when its own bin, or the bin of a destination keypoint passing its gate, lies within 1e-7 of a rounding
    boundary -- there the device's atan2 may legitimately round the other way.  Also reports how many matches differ
    from the exhaustive kernel."""
    differ = total = 0
    for seed in (32, 34):
        views, poses, cam = scene(seed)
        feats = [eng.upload_features(v["xy"], v["desc"], *cam) for v in views]
        pairs = [(0, 1), (1, 0), (0, 2), (2, 1)]
        rt = np.array([np.r_[rel_pose(poses, s, d)[0].ravel(), rel_pose(poses, s, d)[1]] for s, d in pairs])
        got = eng.guided_match_batch(feats, pairs, rt, max_n=max_n, n_bins=45)
        exh = eng.guided_match_batch(feats, pairs, rt, max_n=0, n_bins=0)
        for (s, d), (gi, gj, gr), (xi, xj, xr) in zip(pairs, got, exh):
            oi, oj, orr, frag = oracle_matches_binned(views, poses, cam, s, d, 0)
            assert len(oi) > 300
            if frag.any():   # compare only the rows outside the don't-care band
                keep_o = ~frag[oi].astype(bool)
                oi, oj, orr = oi[keep_o], oj[keep_o], orr[keep_o]
            if max_n and len(oi) > max_n:
                order = np.lexsort((np.arange(len(oi)), orr))[:max_n]
                oi, oj, orr = oi[order], oj[order], orr[order]
            if frag.any():
                keep_g = ~frag[gi].astype(bool)
                gi, gj, gr = gi[keep_g], gj[keep_g], gr[keep_g]
                if max_n:      # the cut may have been taken over a slightly different set: compare the common prefix
                    n = min(len(gi), len(oi)) - int(frag.sum())
                    gi, gj, gr, oi, oj, orr = gi[:n], gj[:n], gr[:n], oi[:n], oj[:n], orr[:n]
            assert np.array_equal(gi, oi) and np.array_equal(gj, oj) and np.array_equal(gr, orr), (seed, s, d)
            if not max_n:
                ex = dict(zip(xi.tolist(), zip(xj.tolist(), xr.tolist())))
                bn = dict(zip(gi.tolist(), zip(gj.tolist(), gr.tolist())))
                total += len(ex)
                differ += sum(1 for k in ex if bn.get(k) != ex[k])
    if not max_n:
        print("binned vs exhaustive kernel: %d of %d matches differ (%.2f %%)" % (differ, total, 100.0 * differ / max(total, 1)))
        assert 0 < differ < 0.05 * total


@pytest.mark.gpu
def test_gpu_guided_match_binned_unusual_geometry(eng):
    """The binned scan looks at every (source, destination) pair of a bin in f32 first and runs the exact gate only on the
    survivors: the f32 look must never drop a pair the exact gate would pass.  Geometries that move the operands across
    ranges -- image coordinates 40x larger and 20x smaller than usual (focal length and size scaled along), an epipole
    inside the image (forward motion: bins evenly filled), and an almost pure rotation (tiny baseline: tiny epipolar
    normals) -- against the literal restatement."""
    views, poses, cam = scene(35, 2000, 2000)
    cases = []
    for scale in (40.0, 0.05):
        vs = [dict(v, xy=(v["xy"] * np.float32(scale)).astype(np.float32)) for v in views]
        cases.append((vs, (cam[0] * scale, cam[1] * scale, cam[2] * scale), [rel_pose(poses, 0, 1), rel_pose(poses, 1, 2)], [(0, 1), (1, 2)]))
    fwd = (np.eye(3), np.array([0.02, -0.01, 1.0]))
    tiny = (rel_pose(poses, 0, 2)[0], rel_pose(poses, 0, 2)[1] * 1e-6)
    cases.append((views, cam, [fwd, tiny], [(0, 1), (0, 2)]))
    checked = 0
    for vs, cm, rts, pairs in cases:
        feats = [eng.upload_features(v["xy"], v["desc"], *cm) for v in vs]
        rt = np.array([np.r_[R.ravel(), t] for R, t in rts])
        got = eng.guided_match_batch(feats, pairs, rt, max_n=0, n_bins=45)
        for (s, d), (R, t), (gi, gj, gr) in zip(pairs, rts, got):
            E = np.zeros(9)
            O.lib().pgo_ref_essential_from_pose(O._p(O.f64(R).ravel()), O._p(O.f64(t)), O._p(E))
            k = [cm[0], cm[0], cm[1] / 2.0, cm[2] / 2.0]
            F = O.fundamental_from_essential(E, k, k)
            size = (int(cm[1]), int(cm[2]))
            oi, oj, orr, frag = O.ref_guided_match_binned(F, vs[s]["xy"], vs[d]["xy"], vs[s]["desc"], vs[d]["desc"], size, size)
            keep_o, keep_g = ~frag[oi].astype(bool), ~frag[gi].astype(bool)
            assert np.array_equal(gi[keep_g], oi[keep_o]) and np.array_equal(gj[keep_g], oj[keep_o]) and np.array_equal(gr[keep_g], orr[keep_o]), (cm, s, d)
            checked += len(oi)
    assert checked > 500


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["flat", "flat_fused", "flat_tiny_arena", "flat_cap4", "flat_cap16", "lanes2", "bin_scan"])
def test_gpu_guided_match_crowded_epipolar_lines(eng, variant, monkeypatch):
    """Source keypoints with far more gate-passing candidates than one pass of the tile scan lists (16 per lane): 120
    destination keypoints are moved onto the epipolar line of each of four source keypoints, so those sources are handled
    in several rounds of ascending destination index -- and `second` (the best BEFORE the last improvement, matcher.h:352-371)
    must still come out as the reference's loop over j leaves it.  Every variant of the scan against the literal restatement."""
    if variant.startswith("flat"):   # the default: the pooled scan (PGI_GUIDED_CAP = candidates a source lists per round)
        monkeypatch.delenv("PGI_GUIDED_LANES", raising=False)
        if "cap" in variant:
            monkeypatch.setenv("PGI_GUIDED_CAP", variant.split("cap")[1])
        if "fused" in variant:   # the scan in ONE kernel (the default is two: window + gate pass + dealing, then sums + pick)
            monkeypatch.setenv("PGI_GUIDED_SPLIT", "0")
        if "tiny_arena" in variant:   # most wavefronts' lists do not fit: they are redone in one kernel
            monkeypatch.setenv("PGI_GUIDED_ARENA_WORDS", "700")
    elif variant == "bin_scan":
        monkeypatch.setenv("PGI_GUIDED_ANGLE", "0")
    else:
        monkeypatch.setenv("PGI_GUIDED_LANES", variant[-1])
    views, poses, cam = scene(36, 1500, 1500)
    s, d = 0, 1
    R, t = rel_pose(poses, s, d)
    E = np.zeros(9)
    O.lib().pgo_ref_essential_from_pose(O._p(O.f64(R).ravel()), O._p(O.f64(t)), O._p(E))
    k = [cam[0], cam[0], cam[1] / 2.0, cam[2] / 2.0]
    F = np.asarray(O.fundamental_from_essential(E, k, k)).reshape(3, 3)
    rng = np.random.default_rng(8)
    xy2 = views[d]["xy"].astype(np.float64).copy()
    crowded = rng.choice(len(views[s]["xy"]), 4, replace=False)
    moved = rng.permutation(len(xy2))[:480].reshape(4, 120)
    for src, rows in zip(crowded, moved):
        l = F @ np.r_[views[s]["xy"][src].astype(np.float64), 1.0]   # x2^T F x1 = 0: the line of x1 in image 2
        nrm = np.hypot(l[0], l[1])
        x = rng.uniform(0.05 * cam[1], 0.95 * cam[1], 120)
        y = -(l[0] * x + l[2]) / l[1]
        off = rng.uniform(-0.3, 0.3, 120)
        xy2[rows, 0] = x + off * l[0] / nrm
        xy2[rows, 1] = y + off * l[1] / nrm
    vs = list(views)
    vs[d] = dict(views[d], xy=xy2.astype(np.float32))
    feats = [eng.upload_features(v["xy"], v["desc"], *cam) for v in vs]
    rt = np.array([np.r_[R.ravel(), t]])
    (gi, gj, gr), = eng.guided_match_batch(feats, [(s, d)], rt, max_n=0, n_bins=45)
    size = (int(cam[1]), int(cam[2]))
    oi, oj, orr, frag = O.ref_guided_match_binned(F.ravel(), vs[s]["xy"], vs[d]["xy"], vs[s]["desc"], vs[d]["desc"], size, size)
    keep_o, keep_g = ~frag[oi].astype(bool), ~frag[gi].astype(bool)
    assert np.array_equal(gi[keep_g], oi[keep_o]) and np.array_equal(gj[keep_g], oj[keep_o]) and np.array_equal(gr[keep_g], orr[keep_o])
    # the crowded sources really had crowds (the exhaustive gate, in numpy)
    for src in crowded:
        l = F @ np.r_[vs[s]["xy"][src].astype(np.float64), 1.0]
        dist = np.abs(vs[d]["xy"].astype(np.float64) @ l[:2] + l[2]) / np.hypot(l[0], l[1])
        assert (dist < 0.5).sum() >= 100


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["flat", "flat_fused", "flat_tiny_arena", "flat_cap4", "flat_cap16", "lanes2", "bin_scan"])
def test_gpu_guided_match_binned_degenerate_inputs(eng, variant, monkeypatch):
    """The binned mode on empty images, a zero pose (F = 0: every record is degenerate, every angle NaN, every destination
    keypoint a candidate of every source: the multi-round path on the records every source visits) and a pure-rotation pose
    (E = 0), against the literal restatement."""
    if variant.startswith("flat"):   # the default: the pooled scan (PGI_GUIDED_CAP = candidates a source lists per round)
        monkeypatch.delenv("PGI_GUIDED_LANES", raising=False)
        if "cap" in variant:
            monkeypatch.setenv("PGI_GUIDED_CAP", variant.split("cap")[1])
        if "fused" in variant:   # the scan in ONE kernel (the default is two: window + gate pass + dealing, then sums + pick)
            monkeypatch.setenv("PGI_GUIDED_SPLIT", "0")
        if "tiny_arena" in variant:   # most wavefronts' lists do not fit: they are redone in one kernel
            monkeypatch.setenv("PGI_GUIDED_ARENA_WORDS", "700")
    elif variant == "bin_scan":
        monkeypatch.setenv("PGI_GUIDED_ANGLE", "0")
    else:
        monkeypatch.setenv("PGI_GUIDED_LANES", variant[-1])
    views, poses, cam = scene(33, 300, 300)
    feats = [eng.upload_features(v["xy"], v["desc"], *cam) for v in views]
    feats.append(eng.upload_features(np.zeros((0, 2), np.float32), np.zeros((0, 128), np.float32), *cam))
    pairs = [(0, 3), (3, 0), (0, 1), (1, 2)]
    rt = np.zeros((4, 12))
    rt[3, :9] = np.eye(3).ravel()
    got = eng.guided_match_batch(feats, pairs, rt, max_n=0, n_bins=45)
    assert len(got[0][0]) == 0 and len(got[1][0]) == 0
    size = (int(cam[1]), int(cam[2]))
    for p in (2, 3):
        s, d = pairs[p]
        E = np.zeros(9)
        O.lib().pgo_ref_essential_from_pose(O._p(O.f64(rt[p, :9])), O._p(O.f64(rt[p, 9:])), O._p(E))
        k = [cam[0], cam[0], cam[1] / 2.0, cam[2] / 2.0]
        oi, oj, orr, frag = O.ref_guided_match_binned(O.fundamental_from_essential(E, k, k), views[s]["xy"], views[d]["xy"],
                                                      views[s]["desc"], views[d]["desc"], size, size)
        keep_o, keep_g = ~frag[oi].astype(bool), ~frag[got[p][0]].astype(bool)
        assert np.array_equal(got[p][0][keep_g], oi[keep_o]) and np.array_equal(got[p][1][keep_g], oj[keep_o])
        assert np.array_equal(got[p][2][keep_g], orr[keep_o], equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("max_n", [1, 7, 100, 333, 1024, 1500])
def test_gpu_guided_match_cut_with_tied_ratios(eng, max_n):
    """The top-N cut (pose_graph_builder.h:759-772: smallest adapted ratio first, ties by position) where the cut falls INSIDE
    groups of equal ratios: every source keypoint exists eight times (same position, same descriptor), so the kept matches
    come in groups of eight identical ratios.  Cuts below and above the radix-select kernel's limit of 1024."""
    views, poses, cam = scene(37, 2500, 500)
    v0 = dict(views[0])
    v0["xy"] = np.repeat(views[0]["xy"], 8, axis=0)
    v0["desc"] = np.repeat(views[0]["desc"], 8, axis=0)
    vs = [v0, views[1]]
    feats = [eng.upload_features(v["xy"], v["desc"], *cam) for v in vs]
    R, t = rel_pose(poses, 0, 1)
    (gi, gj, gr), = eng.guided_match_batch(feats, [(0, 1)], np.r_[R.ravel(), t][None], max_n=max_n, n_bins=0)
    oi, oj, orr, _ = oracle_matches(vs, poses, cam, 0, 1, 0)
    assert len(oi) > 1600 and len(np.unique(orr)) <= len(orr) // 8 + 1
    order = np.lexsort((np.arange(len(oi)), orr))[:max_n]
    assert np.array_equal(gi, oi[order]) and np.array_equal(gj, oj[order]) and np.array_equal(gr, orr[order])


@pytest.mark.gpu
def test_gpu_guided_match_binned_images_beyond_the_bin_scan_limit(eng):
    """20 000 keypoints on either side: more than the bin scan's counters hold (PGI_DESC_MAX = 16 384), inside the tile scan's
    16-bit record positions.  Against the literal restatement."""
    views, poses, cam = scene(38, 2000, 1000)
    rng = np.random.default_rng(5)
    big = []
    for v in views[:2]:   # pad with clutter keypoints up to 20 000
        extra = 20000 - len(v["xy"])
        xy = np.vstack([v["xy"], np.c_[rng.uniform(0, cam[1], extra), rng.uniform(0, cam[2], extra)].astype(np.float32)])
        d = rng.random((extra, 128)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        big.append(dict(xy=xy, desc=np.vstack([v["desc"], d.astype(np.float32)])))
    feats = [eng.upload_features(v["xy"], v["desc"], *cam) for v in big]
    R, t = rel_pose(poses, 0, 1)
    (gi, gj, gr), = eng.guided_match_batch(feats, [(0, 1)], np.r_[R.ravel(), t][None], max_n=0, n_bins=45)
    E = np.zeros(9)
    O.lib().pgo_ref_essential_from_pose(O._p(O.f64(R).ravel()), O._p(O.f64(t)), O._p(E))
    k = [cam[0], cam[0], cam[1] / 2.0, cam[2] / 2.0]
    size = (int(cam[1]), int(cam[2]))
    oi, oj, orr, frag = O.ref_guided_match_binned(O.fundamental_from_essential(E, k, k), big[0]["xy"], big[1]["xy"], big[0]["desc"],
                                                  big[1]["desc"], size, size)
    keep_o, keep_g = ~frag[oi].astype(bool), ~frag[gi].astype(bool)
    assert len(oi) > 500
    assert np.array_equal(gi[keep_g], oi[keep_o]) and np.array_equal(gj[keep_g], oj[keep_o]) and np.array_equal(gr[keep_g], orr[keep_o])


@pytest.mark.gpu
@pytest.mark.parametrize("cap_mb", [1, 0])
def test_gpu_guided_arena_is_sized_by_the_sources_and_capped(cap_mb, monkeypatch):
    """the two-kernel scan's list arena.  (i) Its size follows the wavefronts that hold sources -- a batch
    with ONE large view among small ones must not reserve the large view's share for every pair.  (ii) It is capped by what the
    device has free (here: PGI_GUIDED_ARENA_CAP_MB): with 1 MB most wavefronts' lists do not fit and are redone by the
    one-kernel scan, with 0 the call runs the one-kernel scan outright -- never an error, always the uncapped result."""
    import torch
    from pyposegraphbuilder import Engine
    base, poses, cam = scene(41, 10000, 10000)                 # three views of 20 000 keypoints
    sizes = [20000, 300, 400, 500, 350, 250, 450, 380, 320]     # ... cut into one large view among small ones
    views = [dict(xy=base[k % 3]["xy"][:n], desc=base[k % 3]["desc"][:n]) for k, n in enumerate(sizes)]
    sizes = [len(v["xy"]) for v in views]
    pairs = [(k, (k + 1) % len(sizes)) for k in range(len(sizes))] * 8   # 72 pairs, the large view is the source of 8
    rt = np.zeros((len(pairs), 12))
    for p, (s_, d_) in enumerate(pairs):
        R, t = rel_pose(poses, s_ % 3, d_ % 3)
        rt[p, :9], rt[p, 9:] = np.asarray(R).ravel(), t
    monkeypatch.delenv("PGI_GUIDED_LANES", raising=False)
    monkeypatch.delenv("PGI_GUIDED_ARENA_CAP_MB", raising=False)

    def run():
        e = Engine()
        try:
            feats = [e.upload_features(v["xy"], v["desc"], *cam) for v in views]
            torch.cuda.synchronize()
            free0 = torch.cuda.mem_get_info()[0]
            out = e.guided_match_batch(feats, pairs, rt, max_n=0, n_bins=45)
            torch.cuda.synchronize()
            return out, free0 - torch.cuda.mem_get_info()[0]
        finally:
            e.close()
    ref, ws_uncapped = run()
    # (i) sum of ceil(n1 / 64) wavefronts x 2560 words (10 KB), not ceil(max n1 / 64) x pairs x 10 KB (= 230 MB here)
    waves = sum((sizes[s_] + 63) // 64 for s_, _ in pairs)
    worst = ((max(sizes) + 63) // 64) * len(pairs)
    assert waves * 5 < worst
    assert ws_uncapped < waves * 10240 + (96 << 20), (ws_uncapped, waves)
    monkeypatch.setenv("PGI_GUIDED_ARENA_CAP_MB", str(cap_mb))
    got, ws_capped = run()
    assert ws_capped <= ws_uncapped
    assert sum(len(g[0]) for g in ref) > 100
    for g, r in zip(got, ref):
        for a, b in zip(g, r):
            assert np.array_equal(a, b, equal_nan=True)
