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
