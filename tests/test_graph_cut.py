"""Graph-cut local optimisation (pgi_params.lo_graph_cut, the "GC" of GC-RANSAC): the oracle's labelling step.

The neighbourhood graph is a set of chains (rows of a 4-D grid cell in index order), so the minimum s-t cut of the binary
energy is found by a forward / backward sweep (oracle/pgi_oracle.c: pgo_gc_cut).  Here: that sweep against a GENERIC max-flow
(scipy) on the standard graph construction for submodular binary energies; the chain builder against a dictionary model; the
behaviour of the mode on the estimator.  GPU == oracle bit for bit: tests/test_gpu_parity.py."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from pyposegraphbuilder import synthetic as S

NONE = 0xFFFFFFFF
LEVELS, UNARY = 16, 128


def _lib():
    L = O.lib()
    L.pgo_gc_cut.restype = C.c_uint32
    L.pgo_gc_energy.restype = C.c_int64
    L.pgo_gc_cell.restype = C.c_uint32
    L.pgo_gc_cell.argtypes = [C.c_float] * 4
    L.pgo_gc_kernel_level.restype = C.c_uint32
    return L


def sweep(k, prev, lam):
    L = _lib()
    k, prev = np.ascontiguousarray(k, np.uint32), np.ascontiguousarray(prev, np.uint32)
    labels = np.zeros(len(k), np.uint8)
    cnt = L.pgo_gc_cut(O._p(k), O._p(prev), C.c_uint32(len(k)), C.c_uint32(lam), O._p(labels))
    e = L.pgo_gc_energy(O._p(k), O._p(prev), O._p(labels), C.c_uint32(len(k)), C.c_uint32(lam))
    assert cnt == int(labels.sum())
    return labels, int(e)


def energy_py(k, prev, labels, lam):
    e = 0
    for i in range(len(k)):
        e += UNARY * (LEVELS - int(k[i])) if labels[i] else UNARY * int(k[i])
        if prev[i] != NONE:
            p = int(prev[i])
            s = int(k[p]) + int(k[i])
            e += lam * 2 * LEVELS if labels[i] != labels[p] else (lam * (2 * LEVELS - s) if labels[i] else lam * s)
    return e


def min_cut_energy(k, prev, lam):
    """Kolmogorov-Zabih construction: node x = 1 (inlier) on the sink side.  E = const + max-flow."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import maximum_flow
    n = len(k)
    s, t = n, n + 1
    u0 = np.array([UNARY * int(v) for v in k], np.int64)             # cost of label 0 (outlier)
    u1 = np.array([UNARY * (LEVELS - int(v)) for v in k], np.int64)  # cost of label 1 (inlier)
    const = 0
    rows, cols, caps = [], [], []
    for i in range(n):
        if prev[i] == NONE:
            continue
        p = int(prev[i])
        sm = int(k[p]) + int(k[i])
        A, B, Cc, D = lam * sm, lam * 2 * LEVELS, lam * 2 * LEVELS, lam * (2 * LEVELS - sm)   # E(0,0) E(0,1) E(1,0) E(1,1), x = (p, i)
        # E(xp, xi) = A + (Cc - A) xp + (D - Cc) xi + (B + Cc - A - D) (1 - xp) xi
        const += A
        u1[p] += Cc - A
        u1[i] += D - Cc
        w = B + Cc - A - D
        assert w >= 0
        rows.append(p); cols.append(i); caps.append(w)   # cut when p on the source side (0) and i on the sink side (1)
    # unary: make both costs non-negative, then s -> i with cap u1 (paid when i is on the sink side), i -> t with cap u0
    m = np.minimum(u0, u1)
    const += int(m.sum())
    u0, u1 = u0 - m, u1 - m
    for i in range(n):
        if u1[i]:
            rows.append(s); cols.append(i); caps.append(int(u1[i]))
        if u0[i]:
            rows.append(i); cols.append(t); caps.append(int(u0[i]))
    g = csr_matrix((np.array(caps, np.int32), (rows, cols)), shape=(n + 2, n + 2))
    return const + int(maximum_flow(g, s, t).flow_value)


@pytest.mark.parametrize("lam", [0, 3, 9, 40])
def test_sweep_is_the_minimum_cut(lam):
    rng = np.random.default_rng(100 + lam)
    for trial in range(25):
        n = int(rng.integers(1, 120))
        cells = rng.integers(0, max(1, n // int(rng.integers(1, 6))), n)    # chains of random lengths
        last, prev = {}, np.full(n, NONE, np.uint32)
        for i, c in enumerate(cells):
            prev[i] = last.get(int(c), NONE)
            last[int(c)] = i
        k = rng.integers(0, LEVELS + 1, n).astype(np.uint32)
        if trial % 3 == 0:
            k = np.where(rng.random(n) < 0.5, 8, k).astype(np.uint32)      # plenty of ties
        labels, e = sweep(k, prev, lam)
        assert e == energy_py(k, prev, labels, lam)
        assert e == min_cut_energy(k, prev, lam), (trial, n, lam)
        if lam == 0:   # no coherence: every row on its own, inlier iff k > 8 (ties go to "outlier")
            assert np.array_equal(labels, (k > 8).astype(np.uint8))


def test_chains_link_the_rows_of_a_cell_in_index_order():
    L = _lib()
    b = S.make_batch([4242], [700], inlier_ratio=0.6)
    n = 700
    prev = np.zeros(n, np.uint32)
    L.pgo_gc_chains(O._p(b["x1"]), O._p(b["y1"]), O._p(b["x2"]), O._p(b["y2"]), C.c_uint32(n), O._p(prev))
    cell = lambda i: (int(np.floor(np.float32(b["x1"][i]) * np.float32(8))) & 7) | (int(np.floor(np.float32(b["y1"][i]) * np.float32(8))) & 7) << 3 | \
        (int(np.floor(np.float32(b["x2"][i]) * np.float32(8))) & 7) << 6 | (int(np.floor(np.float32(b["y2"][i]) * np.float32(8))) & 7) << 9
    last = {}
    linked = 0
    for i in range(n):
        c = cell(i)
        assert c == L.pgo_gc_cell(C.c_float(b["x1"][i]), C.c_float(b["y1"][i]), C.c_float(b["x2"][i]), C.c_float(b["y2"][i]))
        assert prev[i] == last.get(c, NONE)
        linked += c in last
        last[c] = i
    assert linked > 20   # the neighbourhood is not empty on this kind of data


def test_coherence_pulls_in_and_pushes_out():
    # a chain of good rows with one mediocre row in the middle (k = 7: alone it would be an outlier) -> pulled in;
    # a chain of bad rows with one row just inside the band (k = 10: alone an inlier) -> pushed out
    prev = np.array([NONE, 0, 1, 2, 3], np.uint32)
    good, _ = sweep(np.array([16, 15, 7, 16, 14], np.uint32), prev, 9)
    assert list(good) == [1, 1, 1, 1, 1]
    bad, _ = sweep(np.array([0, 1, 9, 0, 2], np.uint32), prev, 9)
    assert list(bad) == [0, 0, 0, 0, 0]
    alone, _ = sweep(np.array([16, 15, 7, 16, 14], np.uint32), np.full(5, NONE, np.uint32), 9)
    assert list(alone) == [1, 1, 0, 1, 1]


@pytest.mark.parametrize("rho", [0.3, 0.5])
def test_estimator_with_graph_cut_local_optimisation(rho):
    """The mode on the estimator (CPU oracle): same inputs, lo_graph_cut = 9 (lambda 0.14) against the default.  The refit's
    row set changes, scoring / acceptance / stopping do not: accuracy stays where it was (synthetic outliers are uniform, so
    spatial coherence has little to add here; the numbers are printed)."""
    P, N = 96, 600
    ids = np.arange(7700, 7700 + P)
    b = S.make_batch(ids, N, inlier_ratio=rho)
    res = {}
    for lam in (0, 9):
        e, m = O.estimate_pose_batch(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, O.default_params(lo_graph_cut=lam), 3,
                                     pair_id_base=7700, threads=8)
        err = np.array([S.rot_err_deg(e["R"][i].reshape(3, 3), b["R"][i]) if e["status"][i] == 1 else np.inf for i in range(P)])
        res[lam] = (S.auc_at(err, 5.0), float(e["iters"].mean()), float(e["lo_runs"].mean()), e)
    print("rho %.1f: AUC@5 %.4f -> %.4f, hypotheses %.1f -> %.1f, refits %.2f -> %.2f" % (
        rho, res[0][0], res[9][0], res[0][1], res[9][1], res[0][2], res[9][2]))
    assert res[9][0] > res[0][0] - 0.02
    assert not np.array_equal(res[0][3]["E"], res[9][3]["E"])   # the mode does change the refits
