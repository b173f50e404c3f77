"""scheduler_oracle.py -- CPU ORACLE for the wave-scheduled pose-graph run (test infrastructure, NOT product code).

Restates, in plain Python over the other oracles, the loop of PoseGraphBuilder::processImages at correspondence level
(reference: src/pyposegraphbuilder/include/pose_graph_builder.h:391-709) as the product schedules it
(host/pose_graph_builder.cpp PoseGraphBuilder::run):

  candidates in descending similarity, ties by (src, dst)        imagesimilarity_graph.h max-heap order, :420-431
  skipped: similarity below the threshold (the heap never held it), an edge already in the graph in either direction
  (:426-431), fewer than kMinimumPointNumber matches (:550-551)
  waves of `wave_size` pairs; per wave, on the graph committed by the earlier waves:
    findPath (:785-862) for pairs the visibility table connects (:456-457, 568): A* (astar_oracle.py), at most one
    recovered path, its chained pose (graph_traversal.h:290-348) is the pair's pose guess
    InTraversalPoseTester::test (:798-811, graph_traversal.h:194-233): the guess survives with >= 5 rows inside
    (1.5 thr)^2 -- the product screens a whole wave in one launch of its f32 scoring kernel (K2), restated by
    pgo_model_from_essential + pgo_mask_model; with rotation-guided re-estimation (guess_mode 1) there is no screening
    estimatePose (:940-1078) for every pair of the wave: pgo_estimate_pose_batch, seed = seed_base + wave index, pair
    ids = positions in the wave
    commit in wave order: edge (src, dst, pose, inliers / matches) (:645-654), visibility link (:692)

PARITY UNPINNED against the reference binary (it cannot be built here); the reference's own loop is a 20-thread race
(:391-413) whose order is not defined -- waves are this build's deterministic form of it.  Pose algebra is spelled out
operation by operation in the order of host/pose_graph_builder.hpp (SE3d::operator*, inverse,
getEssentialMatrixFromRelativePose), so that chained poses carry the same bits as the host layer's.
"""
import numpy as np

import astar_oracle as AO


def se3_mul(Ra, ta, Rb, tb):
    """(a * b)(x) = a(b(x)); host SE3d::operator* order."""
    R = np.empty((3, 3))
    t = np.empty(3)
    for i in range(3):
        for j in range(3):
            R[i, j] = Ra[i, 0] * Rb[0, j] + Ra[i, 1] * Rb[1, j] + Ra[i, 2] * Rb[2, j]
        t[i] = Ra[i, 0] * tb[0] + Ra[i, 1] * tb[1] + Ra[i, 2] * tb[2] + ta[i]
    return R, t


def se3_inverse(R, t):
    Ri = R.T.copy()
    ti = np.array([-(Ri[i, 0] * t[0] + Ri[i, 1] * t[1] + Ri[i, 2] * t[2]) for i in range(3)])
    return Ri, ti


def recover_path(graph, path):
    """graph_traversal.h:290-348 with the host layer's operation order."""
    R, t = np.eye(3), np.zeros(3)
    for a, b in zip(path[:-1], path[1:]):
        if graph.has_edge(a, b):
            Re, te, _ = graph.edges[(a, b)]
        elif graph.has_edge(b, a):
            Rs, ts, _ = graph.edges[(b, a)]
            Re, te = se3_inverse(Rs, ts)
        else:
            return None
        R, t = se3_mul(Re, te, R, t)
    return R, t


def essential_from_pose(R, t):
    """pose_utils.h:74-86 in the host layer's order: E = [t]x R, each entry a left-to-right sum of three products."""
    tx = np.array([[0.0, -t[2], t[1]], [t[2], 0.0, -t[0]], [-t[1], t[0], 0.0]])
    E = np.empty((3, 3))
    for i in range(3):
        for j in range(3):
            s = 0.0
            for k in range(3):
                s += tx[i, k] * R[k, j]
            E[i, j] = s
    return E


def run_waves(O, pairs, similarity, n_views, wave_size, *, path_finding=True, rotation_guided=False, min_inliers=20,
              min_points=50, similarity_threshold=0.05, weight=0.8, max_depth=5, seed_base=0, graph_cut=0):
    """pairs: list of dict(src, dst, similarity, thr, x1, y1, x2, y2 (float32 arrays)).  similarity(a, b) -> table value.
    O: the oracle binding (tests/oracle_lib.py).  Returns (statistics dict, edges in insertion order)."""
    order = sorted(range(len(pairs)), key=lambda i: (-pairs[i]["similarity"], pairs[i]["src"], pairs[i]["dst"]))
    graph = AO.PoseGraph()
    vis = AO.UnionFind(n_views)
    st = dict(pairs_processed=0, edges_added=0, paths_searched=0, paths_found=0, touched_nodes=0, poses_from_guess=0,
              hypotheses=0, waves=0, quirk_only_guesses=0)
    edges = []
    prm = O.default_params(min_inliers=min_inliers, guess_mode=1 if rotation_guided else 0, lo_graph_cut=graph_cut)
    seed = seed_base

    def flush(wave):
        nonlocal seed
        if not wave:
            return
        n = len(wave)
        guesses = np.zeros((n, 12))
        has = np.zeros(n, np.uint8)
        counts = np.zeros(n, np.int64)
        if path_finding:
            for k, p in enumerate(wave):
                if not vis.has_link(p["src"], p["dst"]):
                    continue
                path, _, touched = AO.astar_get_path(graph, similarity, p["src"], p["dst"], weight, max_depth)
                st["paths_searched"] += 1
                st["touched_nodes"] += touched
                if path is None:
                    continue
                pose = recover_path(graph, path)
                if pose is None:
                    continue
                st["paths_found"] += 1
                guesses[k, :9], guesses[k, 9:] = pose[0].ravel(), pose[1]
                has[k] = 1
        screened = has.copy()
        if path_finding and not rotation_guided:
            for k, p in enumerate(wave):
                if not has[k]:
                    continue
                E = essential_from_pose(guesses[k, :9].reshape(3, 3), guesses[k, 9:])
                _, c = O.mask_model(O.model_from_essential(E), p["x1"], p["y1"], p["x2"], p["y2"], np.float32((1.5 * p["thr"]) ** 2))
                counts[k] = c
                if c < 5:
                    has[k] = 0
        off = np.zeros(n + 1, np.uint64)
        off[1:] = np.cumsum([len(p["x1"]) for p in wave])
        cat = lambda key: np.concatenate([p[key] for p in wave]).astype(np.float32)
        thr = np.array([p["thr"] for p in wave])
        e, _ = O.estimate_pose_batch(cat("x1"), cat("y1"), cat("x2"), cat("y2"), off, thr, prm, seed, pair_id_base=0,
                                     guesses=guesses if has.any() else None, has_guess=has if has.any() else None)
        seed += 1
        for k, p in enumerate(wave):
            st["hypotheses"] += int(e["iters"][k])
            st["poses_from_guess"] += int(e["used_guess"][k])
            if path_finding and not rotation_guided and screened[k] and e["used_guess"][k] and e["status"][k] == 1 and counts[k] < min_inliers:
                st["quirk_only_guesses"] += 1
            if e["status"][k] != 1:
                continue
            R, t = e["R"][k].reshape(3, 3).copy(), e["t"][k].copy()
            score = float(e["n_inl"][k]) / float(max(1, len(p["x1"])))
            if graph.add_edge(p["src"], p["dst"], R, t, score):
                edges.append((p["src"], p["dst"], score, R, t))
                st["edges_added"] += 1
            vis.add_link(p["src"], p["dst"])
        st["pairs_processed"] += n
        st["waves"] += 1

    wave = []
    for i in order:
        p = pairs[i]
        if p["similarity"] < similarity_threshold:
            break
        if graph.has_edge(p["src"], p["dst"]) or graph.has_edge(p["dst"], p["src"]):
            continue
        if len(p["x1"]) < min_points:
            continue
        graph.add_vertex(p["src"])
        graph.add_vertex(p["dst"])
        wave.append(p)
        if len(wave) == wave_size:
            flush(wave)
            wave = []
    flush(wave)
    st["graph_edges"] = len(graph.edges)
    return st, edges
