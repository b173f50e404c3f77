"""rotavg_oracle.py -- CPU ORACLE for rotation averaging (test infrastructure, NOT product code).

The reference contains NO rotation averaging (grep rotation.?averag|IRLS under /root/reference:
no hits) -- BASELINE.json:north_star adds "L1 rotation-averaging IRLS" downstream
of the pose graph, so this oracle restates the published algorithm (Chatterjee & Govindu,
"Efficient and Robust Large-Scale Rotation Averaging", ICCV 2013 / TPAMI 2018) with INDEPENDENT
arithmetic: scipy Rotation log/exp maps and a sparse direct solve (the HIP path uses its own
Rodrigues formulas and preconditioned CG).  PARITY UNPINNED against any reference implementation;
pinned instead by exact recovery on noise-free graphs and by tests/test_rotavg.py.

Conventions (reference pose.h:14, graph_traversal.h:340-344): an edge (src=i, dst=j) carries
R_ij ~ R_j R_i^T with world->camera rotations R_k.

Specification (mirrored by pose-graph-initialization_amd/csrc/pgi_rotavg.hip):
 1. init: maximum-weight spanning forest (Kruskal over edges sorted by (weight desc, index asc)),
    BFS from the smallest vertex id of each component, R_root = I, R_j = R_ij R_i along tree edges.
 2. outer iteration: per edge dR = R_j^T R_ij R_i, omega = log(dR) in R^3.
    weights: L1 phase (first l1_iters iterations) w = weight / max(|omega|, 1e-4);
             IRLS phase w = weight * sigma^2 / (|omega|^2 + sigma^2)^2 * sigma^2   (Geman-McClure, w(0)=weight)
 3. solve min sum_e w_e |d_j - d_i - omega_e|^2 with d_root = 0  (weighted graph Laplacian, 3 axes)
 4. R_k <- R_k exp(d_k); stop when mean |d_k| < tol or after l1_iters + irls_iters iterations.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from scipy.spatial.transform import Rotation


def spanning_forest_init(n_views, src, dst, Rrel, weight):
    order = sorted(range(len(src)), key=lambda e: (-weight[e], e))
    parent = list(range(n_views))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    adj = [[] for _ in range(n_views)]
    for e in order:
        a, b = find(src[e]), find(dst[e])
        if a != b:
            parent[max(a, b)] = min(a, b)
            adj[src[e]].append((dst[e], e, False))
            adj[dst[e]].append((src[e], e, True))
    R = np.tile(np.eye(3), (n_views, 1, 1))
    seen = np.zeros(n_views, bool)
    roots = []
    for r in range(n_views):
        if seen[r]:
            continue
        roots.append(r)
        seen[r] = True
        queue = [r]
        while queue:
            nxt = []
            for u in queue:
                for v, e, inv in sorted(adj[u], key=lambda t: t[1]):
                    if seen[v]:
                        continue
                    seen[v] = True
                    # forward edge (u=src): R_v = R_uv R_u ; reversed (u=dst): R_v = R_vu^T R_u
                    R[v] = (Rrel[e].T if inv else Rrel[e]) @ R[u]
                    nxt.append(v)
            queue = nxt
    return R, np.array(roots)


def rotation_average(n_views, src, dst, Rrel, weight, l1_iters=5, irls_iters=100, sigma_deg=5.0, tol=1e-8):
    src, dst = np.asarray(src, int), np.asarray(dst, int)
    Rrel = np.asarray(Rrel, float).reshape(-1, 3, 3)
    weight = np.asarray(weight, float)
    R, roots = spanning_forest_init(n_views, src, dst, Rrel, weight)
    E = len(src)
    free = np.ones(n_views, bool)
    free[roots] = False
    fidx = -np.ones(n_views, int)
    fidx[free] = np.arange(free.sum())
    sigma = np.deg2rad(sigma_deg)
    iters = 0
    for it in range(l1_iters + irls_iters):
        dR = np.einsum("eji,ejk,ekl->eil", R[dst], Rrel, R[src])
        om = Rotation.from_matrix(dR).as_rotvec()
        nrm = np.linalg.norm(om, axis=1)
        if it < l1_iters:
            w = weight / np.maximum(nrm, 1e-4)
        else:
            w = weight * sigma ** 2 / (nrm ** 2 + sigma ** 2) ** 2 * sigma ** 2
        rows = np.concatenate([np.arange(E), np.arange(E)])
        cols = np.concatenate([dst, src])
        vals = np.concatenate([np.ones(E), -np.ones(E)])
        A = sp.csr_matrix((vals, (rows, cols)), shape=(E, n_views))[:, free]
        W = sp.diags(w)
        L = (A.T @ W @ A).tocsc()
        d = np.zeros((n_views, 3))
        if L.shape[0]:
            rhs = A.T @ (w[:, None] * om)
            d[free] = np.stack([spla.spsolve(L, rhs[:, k]) for k in range(3)], 1).reshape(-1, 3)
        R = np.einsum("kij,kjl->kil", R, Rotation.from_rotvec(d).as_matrix())
        iters = it + 1
        if np.mean(np.linalg.norm(d, axis=1)) < tol:
            break
    return R, iters


def make_graph(n_views, k=6, noise_deg=1.0, outlier_frac=0.1, seed=0, components=1):
    """Random view graph with ground truth: returns (src, dst, Rrel, weight, R_gt)."""
    rng = np.random.default_rng(seed)
    Rgt = Rotation.random(n_views, random_state=seed).as_matrix()
    comp = np.arange(n_views) % components
    edges = set()
    for i in range(n_views):
        same = np.nonzero(comp == comp[i])[0]
        ring = same[(np.searchsorted(same, i) + 1) % len(same)]
        if ring != i:
            edges.add((min(i, ring), max(i, ring)))
        for j in rng.choice(same, size=min(k, len(same)), replace=False):
            if j != i:
                edges.add((min(i, j), max(i, j)))
    edges = sorted(edges)
    src = np.array([e[0] for e in edges])
    dst = np.array([e[1] for e in edges])
    flip = rng.random(len(edges)) < 0.5
    src, dst = np.where(flip, dst, src), np.where(flip, src, dst)
    Rrel = np.einsum("eij,ekj->eik", Rgt[dst], Rgt[src])
    noise = Rotation.from_rotvec(rng.standard_normal((len(edges), 3)) * np.deg2rad(noise_deg) / np.sqrt(3)).as_matrix()
    Rrel = np.einsum("eij,ejk->eik", noise, Rrel)
    out = rng.random(len(edges)) < outlier_frac
    Rrel[out] = Rotation.random(int(out.sum()), random_state=seed + 1).as_matrix()
    weight = np.where(out, rng.uniform(0.1, 0.4, len(edges)), rng.uniform(0.4, 1.0, len(edges)))
    return src, dst, Rrel, weight, Rgt, out


def align_error_deg(R, Rgt, roots_of=None):
    """Per-view angular error after the gauge alignment R_k ~ Rgt_k G (G from view 0 of each component)."""
    G = Rgt[0].T @ R[0]
    d = np.einsum("kij,jl,kml->kim", Rgt, G, R)
    c = (np.trace(d, axis1=1, axis2=2) - 1) / 2
    return np.degrees(np.arccos(np.clip(c, -1, 1)))
