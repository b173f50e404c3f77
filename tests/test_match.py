"""Descriptor matching (feature_utils.h:135-202): the MFMA path against the scalar oracle, bit-exact.

CPU tests pin the oracle's semantics against an independent float64 numpy restatement; the GPU tests call
pgi_desc_prepare / pgi_match_descriptors_batch through the C ABI and demand identical (src, dst, ratio) lists."""
import numpy as np
import pytest

import oracle_lib as O
from pyposegraphbuilder import synthetic as S


def numpy_matches(A, B):
    """Independent restatement in float64 (selection only; distances differ from f32 in the last bits)."""
    D = np.sqrt(np.maximum(((A[:, None, :].astype(np.float64) - B[None, :, :].astype(np.float64)) ** 2).sum(-1), 0))
    o = np.argsort(D, axis=1, kind="stable")
    i = np.arange(len(A))
    j1, d1, d2 = o[:, 0], D[i, o[:, 0]], D[i, o[:, 1]]
    keep = (d1 < 0.9 * d2) & (np.argmin(D, axis=0)[j1] == i)
    return i[keep], j1[keep], (d1 / d2)[keep]


def test_oracle_against_float64_restatement():
    rng = np.random.default_rng(5)
    A, B, truth = S.make_descriptors(rng, 300, 350, overlap=0.7)
    oi, oj, orr = O.match_descriptors(A, B)
    ni, nj, nr = numpy_matches(A, B)
    assert set(zip(oi.tolist(), oj.tolist())) == set(zip(ni.tolist(), nj.tolist()))
    assert np.all(np.diff(orr) >= 0) and len(oi) > 150          # sorted by ratio (:178-180)
    assert (truth[oi] == oj).mean() > 0.99
    assert np.allclose(np.sort(nr), orr, atol=1e-5)


def test_oracle_needs_two_neighbours_each_way():
    rng = np.random.default_rng(6)
    A, B, _ = S.make_descriptors(rng, 40, 1, overlap=1.0)
    assert len(O.match_descriptors(A, B)[0]) == 0               # feature_utils.h:167-168
    assert len(O.match_descriptors(B, A)[0]) == 0


@pytest.fixture(scope="module")
def eng():
    from pyposegraphbuilder import Engine
    e = Engine()
    yield e
    e.close()


def _check(eng, sets, pairs):
    images = [eng.prepare_descriptors(d) for d in sets]
    got = eng.match_descriptors_batch(images, pairs)
    for (s, d), (gi, gj, gr) in zip(pairs, got):
        oi, oj, orr = O.match_descriptors(sets[s], sets[d]) if len(sets[s]) and len(sets[d]) else ([], [], [])
        assert np.array_equal(gi, np.asarray(oi, np.uint32)), (s, d)
        assert np.array_equal(gj, np.asarray(oj, np.uint32)), (s, d)
        assert np.array_equal(gr, np.asarray(orr, np.float64)), (s, d)      # bit-exact ratios
    return got


@pytest.mark.gpu
def test_gpu_match_bit_exact_ragged(eng):
    rng = np.random.default_rng(7)
    A, B, _ = S.make_descriptors(rng, 700, 900, overlap=0.6)
    C_, D_, _ = S.make_descriptors(rng, 129, 64, overlap=0.9)
    E_, F_, _ = S.make_descriptors(rng, 33, 257, overlap=0.5)
    sets = [A, B, C_, D_, E_, F_]
    got = _check(eng, sets, [(0, 1), (1, 0), (2, 3), (3, 2), (4, 5), (5, 4), (0, 5), (2, 1)])
    assert len(got[0][0]) > 300


@pytest.mark.gpu
def test_gpu_match_distance_ties_and_duplicates(eng):
    """Exact duplicates give equal distances: lowest-index tie-breaks on both sides and ratio == 1 rejections."""
    rng = np.random.default_rng(8)
    A, B, _ = S.make_descriptors(rng, 400, 500, overlap=0.8, duplicates=60)
    A2 = A.copy()
    A2[100:140] = A2[0:40]                                       # duplicate queries: only the lower index is mutual-best
    _check(eng, [A, B, A2], [(0, 1), (1, 0), (2, 1), (1, 2), (2, 2)])


@pytest.mark.gpu
def test_gpu_match_tiny_and_empty_sets(eng):
    rng = np.random.default_rng(9)
    A, B, _ = S.make_descriptors(rng, 40, 1, overlap=1.0)
    Z = np.zeros((0, 128), np.float32)
    two, _, _ = S.make_descriptors(rng, 2, 2, overlap=1.0)
    got = _check(eng, [A, B, Z, two], [(0, 1), (1, 0), (0, 2), (2, 0), (3, 3), (3, 0), (0, 3)])
    assert len(got[0][0]) == 0 and len(got[2][0]) == 0 and len(got[4][0]) == 2


@pytest.mark.gpu
def test_gpu_match_restricted_column_pass_extremes(eng):
    """The column-wise direction only visits the columns some ratio-passing row points at.  One batch with the extremes:
    unrelated sets (hardly a row passes: a list of a few columns), an image against itself (every column listed, several row blocks), a
    permuted copy, and ordinary overlap -- with list lengths that are not multiples of the 256-row block."""
    rng = np.random.default_rng(12)
    R1 = np.abs(rng.standard_normal((700, 128))).astype(np.float32)
    R2 = np.abs(rng.standard_normal((900, 128))).astype(np.float32)
    A, B, _ = S.make_descriptors(rng, 1100, 1000, overlap=0.4)
    Ap = A[rng.permutation(len(A))]
    got = _check(eng, [R1, R2, A, B, Ap], [(0, 1), (2, 2), (2, 4), (4, 2), (2, 3), (3, 2), (1, 0), (0, 3)])
    assert len(got[0][0]) <= 3 and len(got[1][0]) == len(A) and len(got[2][0]) == len(A)


@pytest.mark.gpu
def test_gpu_match_full_size_properties(eng):
    """SIFT-sized sets (8000 keypoints, the reference's -maxkp default region): size-independent properties --
    sortedness, injectivity (mutual best), symmetry of the matched set under swapping the images, and agreement
    with the planted correspondences."""
    rng = np.random.default_rng(10)
    A, B, truth = S.make_descriptors(rng, 8000, 7600, overlap=0.5, noise=0.04)
    images = [eng.prepare_descriptors(A), eng.prepare_descriptors(B)]
    (fi, fj, fr), (bi, bj, br) = eng.match_descriptors_batch(images, [(0, 1), (1, 0)])
    assert np.all(np.diff(fr) >= 0) and np.all(fr < 0.9) and len(fi) > 3000
    assert len(np.unique(fi)) == len(fi) and len(np.unique(fj)) == len(fj)
    assert (truth[fi] == fj).mean() > 0.999
    # a match kept in both directions is the same pair; and every forward match is mutual-best, so its reverse
    # can only be missing because of the reverse ratio test
    fwd, bwd = set(zip(fi.tolist(), fj.tolist())), set(zip(bj.tolist(), bi.tolist()))
    assert len(fwd & bwd) > 0.9 * min(len(fwd), len(bwd))
    # exactness on a slice the oracle finishes quickly: the first 256 queries against all of B
    oi, oj, orr = O.match_descriptors(A[:256], B)
    got = eng.match_descriptors_batch([eng.prepare_descriptors(A[:256]), images[1]], [(0, 1)])[0]
    assert np.array_equal(got[0], oi) and np.array_equal(got[1], oj) and np.array_equal(got[2], orr)


@pytest.mark.gpu
@pytest.mark.parametrize("quirk,top_k", [(False, 0), (True, 0), (False, 100)])
def test_gpu_descriptors_to_pose_on_device(eng, quirk, top_k):
    """descriptors -> matches -> createCorrespondenceMatrix -> estimatePose without leaving the device, against the
    same chain through the oracle: identical matches, bit-identical normalised rows and threshold, identical edges."""
    rng = np.random.default_rng(21)
    views, poses, cam = S.make_feature_views(rng, n_views=3)
    pairs = [(0, 1), (0, 2), (1, 2), (2, 0)]
    images = [eng.prepare_descriptors(v["desc"]) for v in views]
    kps = [eng.upload_keypoints(v["xy"], *cam) for v in views]
    raw = eng.match_descriptors_batch(images, pairs, raw=True)
    b = eng.build_correspondences(kps, pairs, raw, thr_px=2.0, top_k=top_k, dst_uses_src_intrinsics=quirk, seed=77)
    edges, masks = eng.estimate_pose_batch(b)
    e = eng.edges_to_numpy(edges)
    off = b["offsets"].cpu().numpy()
    masks = masks.cpu().numpy()
    x1, y1, x2, y2, thr = (b[k].cpu().numpy() for k in ("x1", "y1", "x2", "y2", "thr"))
    rows = []
    for p, (s, d) in enumerate(pairs):
        oi, oj, _ = O.match_descriptors(views[s]["desc"], views[d]["desc"])
        if top_k:
            oi, oj = oi[:top_k], oj[:top_k]
        assert off[p + 1] - off[p] == len(oi)
        c, t = O.ref_normalize_corr(views[s]["xy"], views[d]["xy"], oi, oj, cam, cam, quirk, 2.0)
        c = c.astype(np.float32)
        sl = slice(off[p], off[p + 1])
        assert np.array_equal(np.stack([x1[sl], y1[sl], x2[sl], y2[sl]], 1), c) and thr[p] == t
        rows.append(c)
    n = int(off[-1])
    c = np.concatenate(rows)
    oe, om = O.estimate_pose_batch(c[:, 0].copy(), c[:, 1].copy(), c[:, 2].copy(), c[:, 3].copy(), off.astype(np.uint64),
                                   thr[:len(pairs)], O.default_params(), seed=77)
    for k in ("status", "n_inl", "score", "iters", "E", "R", "t"):
        assert np.array_equal(e[k], oe[k]), k
    assert np.array_equal(masks[:n], om)
    if not top_k:
        for p, (s, d) in enumerate(pairs):
            R_rel = poses[d][0] @ poses[s][0].T
            assert e["status"][p] == 1 and S.rot_err_deg(e["R"][p].reshape(3, 3), R_rel) < 0.5


@pytest.mark.gpu
def test_gpu_screened_matcher_stress_inputs(eng):
    """Inputs chosen to stress the f16 screen's certificate: clustered descriptors (dozens of columns inside the
    error window), raw integer SIFT (norm ~512), signed entries, zero rows, and norms beyond the f16 range (rows are
    then re-scanned exactly).  The result must stay bit-identical to the oracle in every case."""
    import ctypes as C
    rng = np.random.default_rng(77)
    base, _, _ = S.make_descriptors(rng, 40, 2, overlap=0.0)
    clustered = (base[rng.integers(0, 40, 500)] + 2e-4 * rng.standard_normal((500, 128))).astype(np.float32)
    clustered2 = (base[rng.integers(0, 40, 450)] + 2e-4 * rng.standard_normal((450, 128))).astype(np.float32)
    raw_a = np.floor(np.abs(rng.standard_normal((400, 128))) * 60).clip(0, 255).astype(np.float32)
    raw_b = raw_a[rng.permutation(400)] + rng.integers(-3, 4, (400, 128)).astype(np.float32)
    raw_b = raw_b.clip(0, 255).astype(np.float32)
    signed_a = rng.standard_normal((300, 128)).astype(np.float32)
    signed_b = (signed_a[rng.permutation(300)] + 0.05 * rng.standard_normal((300, 128))).astype(np.float32)
    signed_b[:20] = 0.0                                        # zero descriptors
    huge_a = (signed_a * 3e4).astype(np.float32)                 # squared norms ~1e11: outside the f16 range
    huge_b = (signed_b * 3e4).astype(np.float32)
    sets = [clustered, clustered2, raw_a, raw_b, signed_a, signed_b, huge_a, huge_b]
    pairs = [(0, 1), (1, 0), (2, 3), (3, 2), (4, 5), (5, 4), (6, 7), (7, 6), (0, 5)]
    _check(eng, sets, pairs)
    eng._lib.pgi_internal_match_flagged.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    f, b = C.c_uint64(0), C.c_uint64(0)
    assert eng._lib.pgi_internal_match_flagged(eng._ctx, C.byref(f), C.byref(b)) == 0
    assert f.value > 500 and b.value > 500                      # the clustered and out-of-range sets did take the exact path
    # ordinary data: the fallback stays rare
    A, B, _ = S.make_descriptors(rng, 3000, 3000, overlap=0.6)
    eng.match_descriptors_batch([eng.prepare_descriptors(A), eng.prepare_descriptors(B)], [(0, 1)], raw=True)
    eng._lib.pgi_internal_match_flagged(eng._ctx, C.byref(f), C.byref(b))
    assert f.value + b.value < 0.03 * 6000, (f.value, b.value)
