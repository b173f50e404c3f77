"""pipeline_oracle.py -- CPU ORACLE for the feature-level run (test infrastructure, NOT product code).

Restates, in plain Python over the other oracles, the loop body of PoseGraphBuilder::processImages on in-memory features
(reference: src/pyposegraphbuilder/include/pose_graph_builder.h:391-709) as the product schedules it
(host/pose_graph_builder.cpp PoseGraphBuilder::processFeatures):

  candidates in descending similarity, ties by (src, dst); skipped below the threshold / when the graph holds the edge in
  either direction (:420-431); waves of `wave_size` pairs; per wave, on the state committed by the earlier waves:
    (1) quick matching (:493-518): for a pair the visibility table connects, Tracklets::getCorrespondences
        (tracklets_oracle.py) with kMaximumTrackletNumber; at least kMinimumInlierNumber of them make the pair "quick"
    (2) descriptor matching for the others (:521-546): pgo_match_descriptors
    batch slots: descriptor-matched pairs first, quick pairs after them, each group in wave order (the product's layout
    of a wave; the slot is the pair id of the sampler, the wave index its seed)
        fewer than kMinimumPointNumber matches: the pair is skipped (:550-551) -- an empty slot
    (3) createCorrespondenceMatrix (:553-565): pgo_ref_normalize_corr, rows in f32
    (4) findPath (:568-599, :785-862) for visible, non-skipped pairs: A* (astar_oracle.py) + the chained pose with the host
        layer's operation order (scheduler_oracle.py); InTraversalPoseTester (:798-811): >= 5 rows inside (1.5 thr)^2 under
        the product's f32 scoring (pgo_model_from_essential + pgo_mask_model); none with rotation-guided guesses
    (5) estimatePose (:616-627): pgo_estimate_pose_batch over the slots
    (6) commit in wave order (:645-654, :692): edge (src, dst, pose, inliers / matches), visibility link
    (7) tracklets in wave order (:657-686, :702-709): a quick pair adds the matches of the guided matcher run on its new
        pose -- HashingBasedMatcherWithPose<false, 45> (:738; pgo_ref_guided_match_binned), the
        kMaximumPointNumberForEpipolarHashing smallest adapted ratios, ties by position (:759-772) -- a descriptor-matched
        pair its matches under the estimator's inlier mask

PARITY UNPINNED against the reference binary (it cannot be built here); the reference's own loop is a 20-thread race
(:391-413) whose order is not defined -- waves are this build's deterministic form of it.
"""
import numpy as np

import astar_oracle as AO
import scheduler_oracle as SO
import tracklets_oracle as TO


def run_features(O, views, cam, pairs, similarity, wave_size, *, path_finding=True, hashing=True, rotation_guided=False,
                 min_inliers=20, min_points=50, max_tracklets=5000, max_guided=100, thr_px=0.75, similarity_threshold=0.05,
                 weight=0.8, max_depth=5, n_bins=45, trace=None, progressive=True):
    """views[v] = dict(xy f32 [K,2], desc f32 [K,128]); cam = (focal, width, height) shared by the views; pairs = [(src, dst,
    similarity)]; similarity(a, b) -> table value.  O: the oracle binding (tests/oracle_lib.py).
    trace (optional list) receives (wave, (src, dst), quick, matches, skipped) per slot.
    Returns (statistics dict, edges {(src, dst): (score, R, t)}, number of guided-matching rows in the don't-care band)."""
    V = len(views)
    cand = sorted(pairs, key=lambda p: (-p[2], p[0], p[1]))
    graph = AO.PoseGraph()
    vis = AO.UnionFind(V)
    tracks = TO.Tracklets()
    st = dict(pairs_processed=0, edges_added=0, paths_searched=0, paths_found=0, touched_nodes=0, poses_from_guess=0, hypotheses=0,
              waves=0, matching_runs=0, quick_matching_runs=0, guided_matching_runs=0, guided_matches_added=0, too_few_matches=0,
              quirk_only_guesses=0)
    # (progressive sampling over the matcher's ratio-sorted rows: processFeatures switches it on, PoseGraphBuilder::setProgressiveSampling)
    prm = O.default_params(min_inliers=min_inliers, guess_mode=1 if rotation_guided else 0, sampler=1 if progressive else 0)
    k_cam = [cam[0], cam[0], cam[1] / 2.0, cam[2] / 2.0]
    size = (int(cam[1]), int(cam[2]))
    fragile = 0
    cursor = 0
    seed = 0
    while True:
        wave = []
        while cursor < len(cand) and len(wave) < wave_size:
            s, d, sim = cand[cursor]
            if sim < similarity_threshold:
                cursor = len(cand)
                break
            cursor += 1
            if graph.has_edge(s, d) or graph.has_edge(d, s):
                continue
            wave.append((int(s), int(d)))
        if not wave:
            break
        P = len(wave)
        visible = [vis.has_link(s, d) for s, d in wave]
        matches = [None] * P
        quick = [False] * P
        for i, (s, d) in enumerate(wave):            # (1)
            if hashing and visible[i]:
                m = tracks.get_correspondences(s, d, max_tracklets)
                if len(m) >= min_inliers:
                    quick[i] = True
                    matches[i] = (np.array([a for a, _ in m], np.uint32), np.array([b for _, b in m], np.uint32))
                    st["quick_matching_runs"] += 1
        order = [i for i in range(P) if not quick[i]] + [i for i in range(P) if quick[i]]
        for i in order:                              # (2)
            if quick[i]:
                continue
            s, d = wave[i]
            oi, oj, _ = O.match_descriptors(views[s]["desc"], views[d]["desc"])
            matches[i] = (oi.copy(), oj.copy())
            st["matching_runs"] += 1
        count = [len(matches[i][0]) for i in range(P)]
        skipped = [count[i] < min_points for i in order]          # per slot
        st["too_few_matches"] += sum(skipped)
        if trace is not None:
            trace.extend((st["waves"], wave[i], quick[i], count[i], skipped[k]) for k, i in enumerate(order))
        rows, thr = [], np.zeros(P)
        for k, i in enumerate(order):                # (3)
            s, d = wave[i]
            ms, md = matches[i] if not skipped[k] else (np.zeros(0, np.uint32), np.zeros(0, np.uint32))
            c, thr[k] = O.ref_normalize_corr(views[s]["xy"], views[d]["xy"], ms, md, cam, cam, False, thr_px)
            rows.append(c.astype(np.float32))
        guesses = np.zeros((P, 12))
        has = np.zeros(P, np.uint8)
        if path_finding:                             # (4)
            for k, i in enumerate(order):
                if not visible[i] or skipped[k]:
                    continue
                s, d = wave[i]
                path, _, touched = AO.astar_get_path(graph, similarity, s, d, weight, max_depth)
                st["paths_searched"] += 1
                st["touched_nodes"] += touched
                if path is None:
                    continue
                pose = SO.recover_path(graph, path)
                if pose is None:
                    continue
                st["paths_found"] += 1
                guesses[k, :9], guesses[k, 9:] = pose[0].ravel(), pose[1]
                has[k] = 1
        screened = has.copy()
        counts = np.zeros(P, np.int64)
        if path_finding and not rotation_guided:
            for k in range(P):
                if not has[k]:
                    continue
                E = SO.essential_from_pose(guesses[k, :9].reshape(3, 3), guesses[k, 9:])
                c = rows[k]
                _, counts[k] = O.mask_model(O.model_from_essential(E), c[:, 0].copy(), c[:, 1].copy(), c[:, 2].copy(), c[:, 3].copy(),
                                            np.float32((1.5 * thr[k]) ** 2))
                if counts[k] < 5:
                    has[k] = 0
        off = np.zeros(P + 1, np.uint64)             # (5)
        off[1:] = np.cumsum([len(c) for c in rows])
        cat = np.concatenate(rows) if off[-1] else np.zeros((0, 4), np.float32)
        e, masks = O.estimate_pose_batch(cat[:, 0].copy(), cat[:, 1].copy(), cat[:, 2].copy(), cat[:, 3].copy(), off, thr, prm, seed,
                                         pair_id_base=0, guesses=guesses if has.any() else None, has_guess=has if has.any() else None)
        seed += 1
        slot_of = {i: k for k, i in enumerate(order)}
        committed = []
        for i in range(P):                           # (6)
            k = slot_of[i]
            st["pairs_processed"] += 1
            if skipped[k]:
                continue
            st["hypotheses"] += int(e["iters"][k])
            st["poses_from_guess"] += int(e["used_guess"][k])
            if path_finding and not rotation_guided and screened[k] and e["used_guess"][k] and e["status"][k] == 1 and counts[k] < min_inliers:
                st["quirk_only_guesses"] += 1
            if e["status"][k] != 1:
                continue
            s, d = wave[i]
            R, t = e["R"][k].reshape(3, 3).copy(), e["t"][k].copy()
            graph.add_vertex(s)
            graph.add_vertex(d)
            if graph.add_edge(s, d, R, t, float(e["n_inl"][k]) / float(count[i])):
                st["edges_added"] += 1
            vis.add_link(s, d)
            committed.append((i, k, R, t))
        for i, k, R, t in committed if hashing else []:   # (7)
            s, d = wave[i]
            if quick[i]:
                E = np.zeros(9)
                O.lib().pgo_ref_essential_from_pose(O._p(O.f64(R).ravel()), O._p(O.f64(t)), O._p(E))
                F = O.fundamental_from_essential(E, k_cam, k_cam)
                oi, oj, orr, frag = O.ref_guided_match_binned(F, views[s]["xy"], views[d]["xy"], views[s]["desc"], views[d]["desc"],
                                                              size, size, n_bins)
                fragile += int(frag.sum())
                if len(oi) > max_guided:
                    keep = np.lexsort((np.arange(len(oi)), orr))[:max_guided]
                    oi, oj = oi[keep], oj[keep]
                st["guided_matching_runs"] += 1
                st["guided_matches_added"] += len(oi)
                tracks.add(s, d, list(zip(oi.tolist(), oj.tolist())), [1] * len(oi))
            else:
                a, z = int(off[k]), int(off[k + 1])
                tracks.add(s, d, list(zip(matches[i][0].tolist(), matches[i][1].tolist())), masks[a:z].tolist())
        st["waves"] += 1
    st["graph_edges"] = len(graph.edges)
    st["track_number"] = len(tracks.tracks)
    return st, {key: (v[2], v[0], v[1]) for key, v in graph.edges.items()}, fragile
