"""Synthetic two-view correspondences (SURVEY.md §8d / BASELINE.md §2).

Neither 1DSfM data nor OpenCV exists on either box, so every test and bench
input is generated here.  Pair p uses its own Philox stream keyed by
``seed_base + p`` so any subset of pairs (a rank's shard, a small parity case)
is bit-identical to the same pairs inside the full configuration.

Convention: x_dst ~ R x_src + t  (T_dst_src; reference pose.h:14, pose_utils.h:74-86).
Coordinates are normalised (pixel / focal, principal point removed) exactly as
createCorrespondenceMatrix hands them to estimatePose (pose_graph_builder.h:917-931).
"""
import numpy as np

SEED_BASE = 0x5EED0000
FOCAL_PX = 1000.0
DEFAULT_THR_PX = 0.75


def rodrigues(axis, angle):
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def make_pair(pair_id, n, inlier_ratio=0.5, noise_px=0.25, max_angle_deg=30.0,
              seed_base=SEED_BASE):
    """Returns dict(x1,y1,x2,y2 float32[n], R[3,3], t[3], inlier bool[n])."""
    rng = np.random.Generator(np.random.Philox(key=seed_base + int(pair_id)))
    axis = rng.standard_normal(3)
    angle = np.deg2rad(rng.uniform(0.0, max_angle_deg))
    t = rng.standard_normal(3)
    t /= np.linalg.norm(t)
    R = rodrigues(axis, angle)
    n_in = int(round(n * inlier_ratio))
    p1 = np.zeros((0, 2))
    p2 = np.zeros((0, 2))
    tries = 0
    while len(p1) < n_in:
        m = 4 * max(n_in, 16)
        z = rng.uniform(2.0, 8.0, m)
        X = np.stack([rng.uniform(-0.45, 0.45, m) * z, rng.uniform(-0.45, 0.45, m) * z, z], 1)
        Y = X @ R.T + t
        ok = (Y[:, 2] > 0.5)
        q = Y[:, :2] / np.where(ok, Y[:, 2], 1.0)[:, None]
        ok &= (np.abs(q) < 0.5).all(1)
        p1 = np.concatenate([p1, (X[:, :2] / X[:, 2:3])[ok]])
        p2 = np.concatenate([p2, q[ok]])
        tries += 1
        if tries % 8 == 0:  # views barely overlap: halve the rotation, keep the stream
            angle *= 0.5
            R = rodrigues(axis, angle)
            p1 = np.zeros((0, 2))
            p2 = np.zeros((0, 2))
    p1, p2 = p1[:n_in], p2[:n_in]
    n_out = n - n_in
    o1 = rng.uniform(-0.45, 0.45, (n_out, 2))
    o2 = rng.uniform(-0.5, 0.5, (n_out, 2))
    c = np.concatenate([np.concatenate([p1, p2], 1), np.concatenate([o1, o2], 1)])
    c += rng.standard_normal(c.shape) * (noise_px / FOCAL_PX)
    inl = np.zeros(n, bool)
    inl[:n_in] = True
    perm = rng.permutation(n)
    c, inl = c[perm].astype(np.float32), inl[perm]
    return dict(x1=c[:, 0].copy(), y1=c[:, 1].copy(), x2=c[:, 2].copy(), y2=c[:, 3].copy(),
                R=R, t=t, inlier=inl)


def make_batch(pair_ids, n, inlier_ratio=0.5, noise_px=0.25, max_angle_deg=30.0,
               seed_base=SEED_BASE):
    """Flattened (pair, corr) SoA.  ``n`` is an int or a per-pair sequence."""
    pair_ids = np.asarray(pair_ids, np.int64)
    ns = np.broadcast_to(np.asarray(n, np.int64), pair_ids.shape)
    off = np.zeros(len(pair_ids) + 1, np.uint64)
    off[1:] = np.cumsum(ns)
    tot = int(off[-1])
    out = dict(x1=np.empty(tot, np.float32), y1=np.empty(tot, np.float32),
               x2=np.empty(tot, np.float32), y2=np.empty(tot, np.float32),
               inlier=np.empty(tot, bool), offsets=off, pair_ids=pair_ids.astype(np.uint64),
               R=np.empty((len(pair_ids), 3, 3)), t=np.empty((len(pair_ids), 3)))
    for i, (p, k) in enumerate(zip(pair_ids, ns)):
        d = make_pair(p, int(k), inlier_ratio, noise_px, max_angle_deg, seed_base)
        a, b = int(off[i]), int(off[i + 1])
        for key in ("x1", "y1", "x2", "y2", "inlier"):
            out[key][a:b] = d[key]
        out["R"][i], out["t"][i] = d["R"], d["t"]
    return out


def ragged_sizes(pair_ids, lo=50, hi=4000, seed_base=SEED_BASE):
    """N ~ U{lo..hi} per pair, keyed by pair id (config 2's ragged variant)."""
    return np.array([np.random.Generator(np.random.Philox(key=seed_base ^ 0xA5A5 ^ (int(p) << 20)))
                     .integers(lo, hi + 1) for p in pair_ids], np.int64)


def rot_err_deg(R_est, R_gt):
    c = (np.trace(R_est @ R_gt.T) - 1.0) / 2.0
    return np.degrees(np.arccos(np.clip(c, -1.0, 1.0)))


def auc_at(errors_deg, limit=5.0):
    """AUC@limit of the exact empirical recall curve (failures = inf)."""
    e = np.sort(np.asarray(errors_deg, float))
    e = e[e < limit]
    n = len(errors_deg)
    if n == 0:
        return 0.0
    # recall(theta) steps by 1/n at each error; integral = sum (limit - e_i) / n
    return float(np.sum(limit - e) / (n * limit))


# ---- scene-graph surrogates for BASELINE configs 3/4 (1DSfM data is not available on either box) -----------
def ratio_sorted(batch, seed=77):
    """The rows of every pair of a make_batch-like SoA in the order the reference's matcher hands them over: ascending
    second-nearest-neighbour ratio (feature_utils.h:184-186) -- modelled as a quality score that correlates with being an
    inlier (inliers uniform in [0, 0.8), outliers in [0.2, 1.0); lower = better).  Returns a new dict (same offsets, poses);
    what pgi_params.sampler = 1 (progressive sampling) is meant for."""
    rng = np.random.default_rng(seed)
    out = dict(batch)
    off = np.asarray(batch["offsets"], np.int64)
    n = len(batch["x1"])
    quality = np.where(batch["inlier"], rng.random(n) * 0.8, 0.2 + rng.random(n) * 0.8)
    pair_of = np.repeat(np.arange(len(off) - 1), np.diff(off))
    order = np.lexsort((quality, pair_of))          # by pair, then by quality: a stable per-pair sort
    for k in ("x1", "y1", "x2", "y2", "inlier"):
        out[k] = np.ascontiguousarray(batch[k][order])
    return out


def make_pair_from_pose(pair_id, n, R, t, inlier_ratio=0.5, noise_px=0.25, seed_base=SEED_BASE):
    """Like make_pair but for a GIVEN relative pose (x_dst ~ R x_src + t, |t| = 1)."""
    rng = np.random.Generator(np.random.Philox(key=seed_base ^ 0x5CE7E ^ (int(pair_id) << 8)))
    n_in = int(round(n * inlier_ratio))
    p1 = np.zeros((0, 2))
    p2 = np.zeros((0, 2))
    tries = 0
    lim = 0.5
    while len(p1) < n_in:
        m = 4 * max(n_in, 16)
        z = rng.uniform(2.0, 8.0, m)
        X = np.stack([rng.uniform(-0.45, 0.45, m) * z, rng.uniform(-0.45, 0.45, m) * z, z], 1)
        Y = X @ R.T + t
        ok = Y[:, 2] > 0.5
        q = Y[:, :2] / np.where(ok, Y[:, 2], 1.0)[:, None]
        ok &= (np.abs(q) < lim).all(1)
        p1 = np.concatenate([p1, (X[:, :2] / X[:, 2:3])[ok]])
        p2 = np.concatenate([p2, q[ok]])
        tries += 1
        if tries % 8 == 0:
            lim *= 1.5  # barely overlapping views: widen the destination field of view
    p1, p2 = p1[:n_in], p2[:n_in]
    n_out = n - n_in
    c = np.concatenate([np.concatenate([p1, p2], 1),
                        np.concatenate([rng.uniform(-0.45, 0.45, (n_out, 2)), rng.uniform(-0.5, 0.5, (n_out, 2))], 1)])
    c += rng.standard_normal(c.shape) * (noise_px / FOCAL_PX)
    inl = np.zeros(n, bool)
    inl[:n_in] = True
    perm = rng.permutation(n)
    c, inl = c[perm].astype(np.float32), inl[perm]
    return dict(x1=c[:, 0].copy(), y1=c[:, 1].copy(), x2=c[:, 2].copy(), y2=c[:, 3].copy(), inlier=inl)


def make_scene_graph(n_views, k=8, seed=0, median_corr=600, min_corr=60, max_corr=4000, inlier_lo=0.35,
                     inlier_hi=0.8, outlier_pair_frac=0.05, ring=0):
    """Cameras on a jittered ring looking at the scene centre; candidate pairs = k nearest views (+ with ring = r the
    pairs (i, i+1) ... (i, i+r) along the ring: nearest-neighbour sets alone leave gaps between clusters of views, which
    disconnects large graphs).

    Returns dict(R_gt[V,3,3] world->camera, pairs=(src,dst)[E,2], sizes[E], batch=<make_batch-like SoA>,
    wrong[E] bool (pairs whose correspondences are all outliers: a wrongly retrieved image pair))."""
    rng = np.random.Generator(np.random.Philox(key=SEED_BASE ^ 0xC0DE ^ seed))
    ang = np.sort(rng.uniform(0, 2 * np.pi, n_views))
    radius = rng.uniform(9.0, 11.0, n_views)
    C = np.stack([radius * np.cos(ang), rng.uniform(-0.5, 0.5, n_views), radius * np.sin(ang)], 1)  # centres
    R_gt = np.empty((n_views, 3, 3))
    for i in range(n_views):
        zc = -C[i] / np.linalg.norm(C[i])
        zc = rodrigues(rng.standard_normal(3), np.deg2rad(rng.uniform(0, 6.0))) @ zc  # pointing jitter
        xc = np.cross([0.0, 1.0, 0.0], zc)
        xc /= np.linalg.norm(xc)
        yc = np.cross(zc, xc)
        R_gt[i] = np.stack([xc, yc, zc])  # rows = camera axes in world: x_cam = R (X - C)
    pairs = set()
    for i in range(n_views):
        d = np.abs(np.angle(np.exp(1j * (ang - ang[i]))))
        for j in np.argsort(d)[1:k + 1]:
            pairs.add((min(i, int(j)), max(i, int(j))))
    for r in range(1, ring + 1):
        for i in range(n_views):
            j = (i + r) % n_views
            if i != j:
                pairs.add((min(i, j), max(i, j)))
    pairs = np.array(sorted(pairs), np.int64)
    E = len(pairs)
    sizes = np.clip(rng.lognormal(np.log(median_corr), 0.6, E), min_corr, max_corr).astype(np.int64)
    wrong = rng.random(E) < outlier_pair_frac
    off = np.zeros(E + 1, np.uint64)
    off[1:] = np.cumsum(sizes)
    tot = int(off[-1])
    b = dict(x1=np.empty(tot, np.float32), y1=np.empty(tot, np.float32), x2=np.empty(tot, np.float32),
             y2=np.empty(tot, np.float32), inlier=np.empty(tot, bool), offsets=off,
             R=np.empty((E, 3, 3)), t=np.empty((E, 3)))
    for e, (i, j) in enumerate(pairs):
        Rij = R_gt[j] @ R_gt[i].T
        tij = R_gt[j] @ (C[i] - C[j])
        tij /= np.linalg.norm(tij)
        rho = 0.0 if wrong[e] else rng.uniform(inlier_lo, inlier_hi)
        d = make_pair_from_pose(seed * 1000003 + e, int(sizes[e]), Rij, tij, inlier_ratio=rho)
        a, z = int(off[e]), int(off[e + 1])
        for key in ("x1", "y1", "x2", "y2", "inlier"):
            b[key][a:z] = d[key]
        b["R"][e], b["t"][e] = Rij, tij
    return dict(R_gt=R_gt, pairs=pairs, sizes=sizes, batch=b, wrong=wrong)


def make_scene_graph_dense(n_views, k=40, seed=0, median_corr=600, min_corr=60, max_corr=8000, inlier_lo=0.35,
                           inlier_hi=0.8, outlier_pair_frac=0.03, noise_px=0.25, block_pairs=4096):
    """SURVEY 8d's configs 3/4/5 surrogate AT ITS STATED DENSITY (V = 340 / V = 5000, candidate pairs = k nearest views in
    view direction with k ~ 40, N per pair ~ covisibility with median ~ 600 and cap 8000): the scene of make_scene_graph
    (cameras on a jittered ring looking at the centre), generated for ~10^5 pairs / ~7 * 10^7 rows in seconds -- every
    step is a numpy operation over a block of pairs instead of a Python loop over pairs.  One Philox stream per block of pairs
    (not per pair: a subset of pairs is taken from the generated arrays, see take_pairs).  Same return value as
    make_scene_graph."""
    rng = np.random.Generator(np.random.Philox(key=SEED_BASE ^ 0xDE45E ^ (seed << 4)))
    ang = np.sort(rng.uniform(0, 2 * np.pi, n_views))
    radius = rng.uniform(9.0, 11.0, n_views)
    C = np.stack([radius * np.cos(ang), rng.uniform(-0.5, 0.5, n_views), radius * np.sin(ang)], 1)
    zc = -C / np.linalg.norm(C, axis=1, keepdims=True)
    jit_axis = rng.standard_normal((n_views, 3))
    jit_axis /= np.linalg.norm(jit_axis, axis=1, keepdims=True)
    jit_ang = np.deg2rad(rng.uniform(0, 6.0, n_views))
    # Rodrigues on all views: z' = z cos a + (k x z) sin a + k (k . z)(1 - cos a)
    zc = (zc * np.cos(jit_ang)[:, None] + np.cross(jit_axis, zc) * np.sin(jit_ang)[:, None]
          + jit_axis * (np.einsum("ij,ij->i", jit_axis, zc) * (1 - np.cos(jit_ang)))[:, None])
    xc = np.cross(np.array([0.0, 1.0, 0.0]), zc)
    xc /= np.linalg.norm(xc, axis=1, keepdims=True)
    yc = np.cross(zc, xc)
    R_gt = np.stack([xc, yc, zc], 1)  # rows = camera axes in world: x_cam = R (X - C)
    # the k nearest views in view direction: the angles are sorted, so they lie among the k ring neighbours on either side
    kk = min(k, n_views - 1)
    steps = np.concatenate([np.arange(-kk, 0), np.arange(1, kk + 1)])
    nb = (np.arange(n_views)[:, None] + steps[None, :]) % n_views
    dang = np.abs(np.angle(np.exp(1j * (ang[nb] - ang[:, None]))))
    order = np.argsort(dang, axis=1, kind="stable")[:, :kk]
    near = np.take_along_axis(nb, order, 1)
    rank = np.broadcast_to(np.arange(kk)[None, :], near.shape)
    i_all = np.repeat(np.arange(n_views), kk)
    j_all = near.ravel()
    keep = i_all != j_all
    lo, hi, rk = np.minimum(i_all, j_all)[keep], np.maximum(i_all, j_all)[keep], rank.ravel()[keep]
    key = lo.astype(np.int64) * n_views + hi
    o = np.lexsort((rk, key))
    key, rk = key[o], rk[o]
    first = np.concatenate([[True], key[1:] != key[:-1]])  # a pair listed from both ends keeps its smaller rank
    key, rk = key[first], rk[first]
    pairs = np.stack([key // n_views, key % n_views], 1).astype(np.int64)
    E = len(pairs)
    # rows per pair ~ covisibility: closer views share more of the scene
    cov = 1.4 - 0.8 * rk / max(kk - 1, 1)
    sizes = np.clip(rng.lognormal(np.log(median_corr), 0.6, E) * cov, min_corr, max_corr).astype(np.int64)
    wrong = rng.random(E) < outlier_pair_frac
    rho = np.where(wrong, 0.0, rng.uniform(inlier_lo, inlier_hi, E))
    Ri, Rj = R_gt[pairs[:, 0]], R_gt[pairs[:, 1]]
    Rij = np.einsum("eij,ekj->eik", Rj, Ri)
    tij = np.einsum("eij,ej->ei", Rj, C[pairs[:, 0]] - C[pairs[:, 1]])
    tij /= np.linalg.norm(tij, axis=1, keepdims=True)
    off = np.zeros(E + 1, np.uint64)
    off[1:] = np.cumsum(sizes)
    tot = int(off[-1])
    b = dict(x1=np.empty(tot, np.float32), y1=np.empty(tot, np.float32), x2=np.empty(tot, np.float32),
             y2=np.empty(tot, np.float32), inlier=np.empty(tot, bool), offsets=off, R=Rij, t=tij)
    def block(e0):  # one Philox stream per block of pairs: the blocks are independent and run on a few threads
        rb = np.random.Generator(np.random.Philox(key=SEED_BASE ^ 0xB10C ^ (seed << 4) ^ (e0 << 24)))
        e1 = min(E, e0 + block_pairs)
        a, z_ = int(off[e0]), int(off[e1])
        n = z_ - a
        pe = np.repeat(np.arange(e0, e1), sizes[e0:e1])
        inl = rb.random(n) < rho[pe]
        c = np.empty((n, 4))
        out_rows = np.nonzero(~inl)[0]
        c[out_rows, :2] = rb.uniform(-0.45, 0.45, (len(out_rows), 2))
        c[out_rows, 2:] = rb.uniform(-0.5, 0.5, (len(out_rows), 2))
        pend = np.nonzero(inl)[0]       # inlier rows still without a point that both views see
        lim = np.full(len(pend), 0.5)
        tries = 0
        while len(pend):
            m = len(pend)
            zz = rb.uniform(2.0, 8.0, m)
            X = np.stack([rb.uniform(-0.45, 0.45, m) * zz, rb.uniform(-0.45, 0.45, m) * zz, zz], 1)
            pp = pe[pend]
            Y = np.einsum("mij,mj->mi", Rij[pp], X) + tij[pp]
            ok = Y[:, 2] > 0.5
            q = Y[:, :2] / np.where(ok, Y[:, 2], 1.0)[:, None]
            ok &= (np.abs(q) < lim[:, None]).all(1)
            done = pend[ok]
            c[done, :2] = X[ok, :2] / X[ok, 2:3]
            c[done, 2:] = q[ok]
            pend, lim = pend[~ok], lim[~ok]
            tries += 1
            if tries % 8 == 0:
                lim = lim * 1.5  # barely overlapping views: widen the destination field of view
        c += rb.standard_normal((n, 4)) * (noise_px / FOCAL_PX)
        c = c.astype(np.float32)
        b["x1"][a:z_], b["y1"][a:z_], b["x2"][a:z_], b["y2"][a:z_] = c[:, 0], c[:, 1], c[:, 2], c[:, 3]
        b["inlier"][a:z_] = inl
    import os
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max(1, min(16, os.cpu_count() or 1))) as ex:
        list(ex.map(block, range(0, E, block_pairs)))
    return dict(R_gt=R_gt, pairs=pairs, sizes=sizes, batch=b, wrong=wrong)


def take_pairs(g, idx):
    """The pairs `idx` of a scene graph as a make_batch-like SoA (a sample for a parity check against the oracle)."""
    b = g["batch"]
    idx = np.asarray(idx, np.int64)
    n = np.diff(b["offsets"].astype(np.int64))[idx]
    off = np.zeros(len(idx) + 1, np.uint64)
    off[1:] = np.cumsum(n)
    rows = np.concatenate([np.arange(int(b["offsets"][e]), int(b["offsets"][e + 1])) for e in idx]) if len(idx) else np.zeros(0, np.int64)
    out = {k_: b[k_][rows] for k_ in ("x1", "y1", "x2", "y2", "inlier")}
    out.update(offsets=off, R=b["R"][idx], t=b["t"][idx])
    return out


def make_descriptors(rng, n_a, n_b, overlap=0.6, noise=0.05, dim=128, duplicates=0):
    """Two RootSIFT-like descriptor sets (non-negative, unit L2 norm): `overlap` of A re-appears in B, perturbed and
    permuted; `duplicates` rows of B are exact copies of other rows (distance ties)."""
    def unit(x):
        x = np.abs(x).astype(np.float32)
        return x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-12).astype(np.float32)
    A = unit(rng.standard_normal((n_a, dim)))
    B = unit(rng.standard_normal((n_b, dim)))
    m = int(min(n_a, n_b) * overlap)
    ia, ib = rng.permutation(n_a)[:m], rng.permutation(n_b)[:m]
    if m:
        B[ib] = unit(A[ia] + noise * rng.standard_normal((m, dim)))
    for _ in range(duplicates):
        if n_b >= 2:
            s, d = rng.integers(0, n_b, 2)
            B[d] = B[s]
    truth = np.full(n_a, -1, np.int64)
    truth[ia] = ib
    return A.astype(np.float32), B.astype(np.float32), truth


def make_feature_views(rng, n_views=3, n_points=1500, n_clutter=500, f=1000.0, w=1600.0, h=1200.0, desc_noise=0.04,
                       px_noise=0.3):
    """A small scene seen by n_views cameras: every view detects a random ~75% of the 3-D points (pixel keypoints +
    a noisy copy of the point's RootSIFT-like descriptor) plus unmatched clutter.  Returns views[v] = dict(xy, desc,
    point_id) and the ground-truth world->camera poses."""
    def unit(x):
        x = np.abs(x).astype(np.float32)
        return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
    X = np.stack([rng.uniform(-2, 2, n_points), rng.uniform(-1.5, 1.5, n_points), rng.uniform(4, 8, n_points)], 1)
    D = unit(rng.standard_normal((n_points, 128)))
    views, poses = [], []
    for v in range(n_views):
        ax = rng.standard_normal(3)
        R = rodrigues(ax / np.linalg.norm(ax), np.radians(rng.uniform(2, 12))) if v else np.eye(3)
        t = rng.uniform(-0.6, 0.6, 3) * np.array([1, 0.5, 0.3]) if v else np.zeros(3)
        Y = X @ R.T + t
        px = np.stack([f * Y[:, 0] / Y[:, 2] + w / 2, f * Y[:, 1] / Y[:, 2] + h / 2], 1)
        vis = (Y[:, 2] > 0.5) & (px[:, 0] > 0) & (px[:, 0] < w) & (px[:, 1] > 0) & (px[:, 1] < h) & (rng.random(n_points) < 0.75)
        ids = np.nonzero(vis)[0]
        xy = px[ids] + px_noise * rng.standard_normal((len(ids), 2))
        desc = unit(D[ids] + desc_noise * rng.standard_normal((len(ids), 128)))
        cxy = np.stack([rng.uniform(0, w, n_clutter), rng.uniform(0, h, n_clutter)], 1)
        cdesc = unit(rng.standard_normal((n_clutter, 128)))
        perm = rng.permutation(len(ids) + n_clutter)
        views.append(dict(xy=np.concatenate([xy, cxy])[perm].astype(np.float32),
                          desc=np.concatenate([desc, cdesc])[perm].astype(np.float32),
                          point_id=np.concatenate([ids, -np.ones(n_clutter, np.int64)])[perm]))
        poses.append((R, t))
    return views, poses, (f, w, h)


def make_feature_scene(n_views=340, keypoints=8000, band=20, seed=5, clutter_frac=0.25, f=1000.0, w=1600.0, h=1200.0,
                       desc_noise=0.012, px_noise=0.3, step=0.3, detect=0.75):
    """BASELINE config 3's surrogate AT FEATURE LEVEL (1DSfM Madrid Metropolis: ~340 images, ~8000 SIFT keypoints each; the
    data set is on neither box): cameras walk along a street (x axis, `step` apart, small pose jitter) past a slab of 3-D
    points (depth 4..8); every view detects `detect` of the points in its frustum (pixel keypoints with noise + a noisy copy
    of the point's RootSIFT-like descriptor) plus unmatched clutter, about `keypoints` rows in all.  Candidate pairs = the
    `band` next views of every view (what a retrieval network's top-k would give, k ~ 2 * band), similarity falling with the
    distance along the walk.

    Returns views[v] = dict(xy f32 [K,2], desc f32 [K,128], point_id), poses[v] = (R, t) world->camera,
    cam = (f, w, h), sim [V,V], pairs = [(i, j, similarity)]."""
    rng = np.random.Generator(np.random.Philox(key=SEED_BASE ^ 0xFEA7 ^ seed))

    def unit(x):
        x = np.abs(x, out=x)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        return x
    n_clutter = int(keypoints * clutter_frac)
    want_visible = (keypoints - n_clutter) / detect          # points inside a frustum
    half = 0.5 * w / f                                         # tan of the horizontal half angle
    area = half * (8.0 ** 2 - 4.0 ** 2)                        # frustum cross-section in the (x, z) plane, z in [4, 8]
    x_lo, x_hi = -8.0 * half - 1.0, (n_views - 1) * step + 8.0 * half + 1.0
    n_points = int(want_visible / area * (x_hi - x_lo) * 4.0)   # uniform over (x, z): the slab is 4 deep
    X = np.stack([rng.uniform(x_lo, x_hi, n_points), rng.uniform(-1.4, 1.4, n_points), rng.uniform(4, 8, n_points)], 1)
    D = unit(rng.standard_normal((n_points, 128), dtype=np.float32))
    order = np.argsort(X[:, 0])
    X, D = X[order], D[order]
    views, poses = [], []
    for v in range(n_views):
        ax = rng.standard_normal(3)
        R = rodrigues(ax / np.linalg.norm(ax), np.radians(rng.uniform(0.5, 6.0)))
        C = np.array([v * step, 0.0, 0.0]) + rng.uniform(-0.1, 0.1, 3) * np.array([1.0, 0.6, 1.0])
        t = -R @ C
        a, z = np.searchsorted(X[:, 0], [C[0] - 8.5 * half - 1.5, C[0] + 8.5 * half + 1.5])
        Y = X[a:z] @ R.T + t
        px = np.stack([f * Y[:, 0] / Y[:, 2] + w / 2, f * Y[:, 1] / Y[:, 2] + h / 2], 1)
        vis = (Y[:, 2] > 0.5) & (px[:, 0] > 0) & (px[:, 0] < w) & (px[:, 1] > 0) & (px[:, 1] < h) & (rng.random(z - a) < detect)
        ids = a + np.nonzero(vis)[0]
        xy = px[ids - a] + px_noise * rng.standard_normal((len(ids), 2))
        desc = unit(D[ids] + np.float32(desc_noise) * rng.standard_normal((len(ids), 128), dtype=np.float32))
        cxy = np.stack([rng.uniform(0, w, n_clutter), rng.uniform(0, h, n_clutter)], 1)
        cdesc = unit(rng.standard_normal((n_clutter, 128), dtype=np.float32))
        perm = rng.permutation(len(ids) + n_clutter)
        views.append(dict(xy=np.concatenate([xy, cxy])[perm].astype(np.float32),
                          desc=np.concatenate([desc, cdesc])[perm],
                          point_id=np.concatenate([ids, -np.ones(n_clutter, np.int64)])[perm]))
        poses.append((R, t))
    sim = np.zeros((n_views, n_views))
    pairs = []
    for i in range(n_views):
        for j in range(i + 1, min(n_views, i + band + 1)):
            s = round(0.95 - 0.6 * (j - i) / (band + 1) + 0.001 * ((3 * i + j) % 7), 3)
            sim[i, j] = sim[j, i] = s
            pairs.append((i, j, s))
    return views, poses, (f, w, h), sim, pairs
