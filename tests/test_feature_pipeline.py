"""processFeatures (host/pose_graph_builder.cpp): the loop body of processImages (pose_graph_builder.h:391-709) on
in-memory features -- descriptor matching / tracklet quick matching -> createCorrespondenceMatrix -> A* guesses ->
estimatePose -> guided matching -> tracklet update -- driven from C++ on the GPU."""
import os
import struct
import subprocess

import numpy as np
import pytest

import scene_drivers as SC
from pyposegraphbuilder import synthetic as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "pose-graph-initialization_amd", "test_pipeline")
WAVE = 8


def make_scene():
    rng = np.random.default_rng(41)
    views, poses, cam = S.make_feature_views(rng, n_views=10, n_points=2500, n_clutter=2500, desc_noise=0.012)
    V = len(views)
    sim = np.zeros((V, V))
    pairs = []
    for i in range(V):
        for j in range(i + 1, V):
            shared = len(set(views[i]["point_id"][views[i]["point_id"] >= 0]) & set(views[j]["point_id"]))
            sim[i, j] = sim[j, i] = round(0.2 + 0.7 * shared / 2500 + 0.001 * ((3 * i + j) % 7), 3)
            pairs.append((i, j, sim[i, j]))
    return views, poses, cam, sim, pairs


def parse(buf, n_modes=4):
    return SC.parse_pipeline(buf, n_modes)


@pytest.mark.gpu
def test_feature_pipeline_three_configurations(tmp_path):
    views, poses, cam, sim, pairs = make_scene()
    V = len(views)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    SC.write_feature_scene(fin, views, cam, sim, pairs, WAVE)
    r = subprocess.run([EXE, fin, fout], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    (st0, e0), (st1, e1), (st2, e2), (st3, e3) = parse(open(fout, "rb").read())
    # tracklets in HBM (mode 2) and in the host store (mode 3) are the same bookkeeping: every counter and edge agrees
    assert st2 == st3 and e2.keys() == e3.keys()
    for key in e2:
        assert e2[key][0] == e3[key][0] and np.array_equal(e2[key][1], e3[key][1]) and np.array_equal(e2[key][2], e3[key][2])
    for mode, (st, edges) in enumerate(((st0, e0), (st1, e1), (st2, e2))):
        assert st[0] == len(pairs) and st[1] == st[8] == len(edges) and st[1] >= 0.95 * len(pairs)
        err, tcos = [], []
        for (s, d), (sc, R, t) in edges.items():
            R_rel = poses[d][0] @ poses[s][0].T
            t_rel = poses[d][1] - R_rel @ poses[s][1]
            err.append(S.rot_err_deg(R, R_rel))
            assert 0 < sc <= 1
            tcos.append(t @ t_rel / np.linalg.norm(t_rel))
        err = np.array(err)
        print("mode %d: rot err median %.3f deg, <0.5 deg %.2f, max %.2f; t cos median %.4f" %
              (mode, np.median(err), np.mean(err < 0.5), err.max(), np.median(tcos)))
        # chained poses accepted through the reference's loose guess test (graph_traversal.h:164) are refitted on
        # a wide inlier band and are less accurate than robust fits: the plain mode carries the tight bound
        assert np.mean(err < 0.5) > (0.95 if mode == 0 else 0.8) and np.median(err) < 0.1
        # short baselines (0.1-0.6 scene units at depth 4-8) make the direction of t far noisier than R
        assert np.median(tcos) > 0.97 and np.mean(np.array(tcos) > 0.8) > 0.9, (np.median(tcos), np.sort(tcos)[:5])
    # plain: every pair is descriptor-matched, nothing else runs
    assert st0[9] == len(pairs) and st0[2] == 0 and st0[10] == 0 and st0[11] == 0 and st0[13] == 0
    # + path finding: later waves get chained poses and need fewer hypotheses
    assert st1[2] > 0 and st1[5] > 0 and st1[6] < st0[6]
    # + epipolar hashing: tracklets replace descriptor matching for connected pairs, guided matching feeds them back
    assert st2[10] > 0 and st2[9] + st2[10] == len(pairs) and st2[9] < st0[9]
    assert st2[11] > 0 and st2[12] > 0 and st2[13] > 0
    print("plain: %d edges %d hyps | A*: %d guesses used, %d hyps | hashing: %d matched + %d quick, %d guided runs (+%d matches), "
          "%d tracks" % (st0[1], st0[6], st1[5], st1[6], st2[9], st2[10], st2[11], st2[12], st2[13]))

    # the plain C++ run equals the same chain driven from Python through the C ABI, wave by wave, bit for bit
    from pyposegraphbuilder import Engine
    eng = Engine(min_inliers=20, sampler=1)   # (processFeatures samples progressively over the matcher's ratio-sorted rows)
    try:
        images = [eng.prepare_descriptors(v["desc"]) for v in views]
        kps = [eng.upload_keypoints(v["xy"], *cam) for v in views]
        order = sorted(pairs, key=lambda p: (-p[2], p[0], p[1]))
        for w in range(0, len(order), WAVE):
            wave = [(i, j) for i, j, _ in order[w:w + WAVE]]
            raw = eng.match_descriptors_batch(images, wave, max_matches=max(len(views[i]["xy"]) for i, _ in wave), raw=True)
            b = eng.build_correspondences(kps, wave, raw, thr_px=0.75, seed=w // WAVE)
            edges, _ = eng.estimate_pose_batch(b)
            e = eng.edges_to_numpy(edges)
            cnt = raw[3].cpu().numpy()
            for p, (i, j) in enumerate(wave):
                if e["status"][p] != 1:
                    assert (i, j) not in e0
                    continue
                sc, R, t = e0[(i, j)]
                assert np.array_equal(R.ravel(), e["R"][p]) and np.array_equal(t, e["t"][p])
                assert sc == e["n_inl"][p] / cnt[p]
    finally:
        eng.close()


@pytest.mark.gpu
def test_config3_from_features_at_full_size(tmp_path):
    """BASELINE config 3 ("1DSfM Madrid Metropolis (~340 imgs) full pose-graph build on 1 GPU") FROM FEATURES at its stated
    size: 340 views x ~8000 keypoints x 128-d descriptors (1.4 GB), the 20 next views of every view as candidates (k ~ 40,
    6590 pairs), path finding and epipolar hashing on, tracklets in HBM -- processImages' loop body
    (pose_graph_builder.h:391-709) with point_track.h:568-711 behind it.  Checked: (i) against the CORRESPONDENCE-level run
    of the same scene (mode 0: every pair descriptor-matched and estimated on its own) -- same edge set up to a handful,
    rotations of common edges agree; (ii) the device tracklet store against the host store: every counter and every edge
    identical; (iii) against the ground truth."""
    views, poses, cam, sim, pairs = S.make_feature_scene(340, 8000, band=20)
    assert len(views) == 340 and 7500 < np.mean([len(v["xy"]) for v in views]) < 8500 and len(pairs) == 6590
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    SC.write_feature_scene(fin, views, cam, sim, pairs, 512)
    del views
    r = subprocess.run([EXE, fin, fout, "0234"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    print(r.stdout)
    (st0, e0), (st2, e2), (st3, e3), (st4, e4) = parse(open(fout, "rb").read(), 4)
    os.remove(fin)
    # (ii) tracklets in HBM == host store, counter for counter and edge for edge
    assert st2 == st3 and e2.keys() == e3.keys()
    for key in e2:
        assert e2[key][0] == e3[key][0] and np.array_equal(e2[key][1], e3[key][1]) and np.array_equal(e2[key][2], e3[key][2])
    k, k4 = dict(zip(SC.PIPELINE_KEYS, st2)), dict(zip(SC.PIPELINE_KEYS, st4))
    assert st0[0] == st2[0] == st4[0] == len(pairs) and st0[9] == len(pairs)
    # the hashing runs replace most descriptor matches by tracklet look-ups and feed guided matches back
    for kk in (k, k4):
        assert kk["quick_matching_runs"] > 0.5 * len(pairs) and kk["matching_runs"] + kk["quick_matching_runs"] == len(pairs)
        assert kk["guided_matching_runs"] > 0 and kk["guided_matches_added"] > 0 and kk["track_number"] > 100000
    # (i) against the correspondence-level run, (iii) against the ground truth
    common = sorted(set(e0) & set(e2) & set(e4))
    assert min(len(e0), len(e2), len(e4)) >= 0.97 * len(pairs) and len(common) >= 0.97 * len(pairs)
    def gt(s, d):
        return poses[d][0] @ poses[s][0].T
    err0 = np.array([S.rot_err_deg(e0[key][1], gt(*key)) for key in common])
    err2 = np.array([S.rot_err_deg(e2[key][1], gt(*key)) for key in common])
    err4 = np.array([S.rot_err_deg(e4[key][1], gt(*key)) for key in common])
    diff4 = np.array([S.rot_err_deg(e0[key][1], e4[key][1]) for key in common])
    for name, err, kk in (("plain (correspondence level)", err0, dict(zip(SC.PIPELINE_KEYS, st0))), ("A* + hashing, reference guesses", err2, k),
                          ("A* + hashing, rotation-guided", err4, k4)):
        print("config 3 from features, %-32s: %d edges, %d guesses used (%d quirk-only), %d hypotheses; rot err median %.3f deg, "
              "< 0.5 deg %.3f, < 5 deg %.3f" % (name, kk["graph_edges"], kk["poses_from_guess"], kk["quirk_only_guesses"], kk["hypotheses"],
                                                np.median(err), np.mean(err < 0.5), np.mean(err < 5.0)))
    assert np.median(err0) < 0.1 and np.mean(err0 < 0.5) > 0.97
    # rotation-guided re-estimation of the chained poses: the correspondence-level answer, pair for pair
    assert np.median(err4) < 0.1 and np.mean(err4 < 0.5) > 0.97 and np.median(diff4) < 0.1 and np.mean(diff4 < 0.5) > 0.97
    assert k4["quirk_only_guesses"] == 0 and k4["poses_from_guess"] > 0.5 * len(pairs)
    # The reference's guess path (pose_graph_builder.h:974-1029), reproduced by default: a chained pose that passes the
    # 5-inlier tester is accepted on the rows inside the UN-squared bound (graph_traversal.h:164 -- a ~33 px band, i.e.
    # nearly every row, mismatches included) and refitted on all of them.  Most such edges are fine, a good tenth is not;
    # the run counts the guesses only the quirk let through.  Documented, measured, and the reason guess_mode = 1 exists.
    assert np.median(err2) < 0.15 and np.mean(err2 < 5.0) > 0.8 and k["quirk_only_guesses"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("rotationGuided", [False, True])
def test_python_run_features_is_the_cpp_pipeline(tmp_path, rotationGuided):
    """pyposegraphbuilder.PoseGraphBuilder.runFeatures (pgih_run_features of libpgi_host.so) against the C++ driver on the same
    scene, path finding + epipolar hashing with the tracklets in HBM: same counters, same edges, every bit."""
    from pyposegraphbuilder.builder import PoseGraphBuilder
    views, poses, cam, sim, pairs = make_scene()
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    SC.write_feature_scene(fin, views, cam, sim, pairs, WAVE)
    mode = "4" if rotationGuided else "2"
    r = subprocess.run([EXE, fin, fout, mode], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    (st, edges), = parse(open(fout, "rb").read(), 1)
    b = PoseGraphBuilder(20, 5000, 5, 100, 20, 50, 100, 0.8, 0.05, 0.75, "", "", "", "", True, True, True)
    try:
        vs = [dict(xy=v["xy"], desc=v["desc"], focal=cam[0], width=cam[1], height=cam[2]) for v in views]
        graph = b.runFeatures(vs, pairs, waveSize=WAVE, rotationGuided=rotationGuided)
        k = dict(zip(SC.PIPELINE_KEYS, st))
        for name in ("pairs_processed", "edges_added", "paths_searched", "paths_found", "touched_nodes", "poses_from_guess", "hypotheses",
                     "waves", "graph_edges", "matching_runs", "quick_matching_runs", "guided_matching_runs", "guided_matches_added",
                     "track_number", "quirk_only_guesses"):
            assert b.statistics[name] == k[name], name
        assert graph.keys() == edges.keys() and len(graph) > 0.9 * len(pairs)
        for key, (sc, R, t) in edges.items():
            assert graph[key]["score"] == sc and np.array_equal(graph[key]["R"], R.reshape(3, 3)) and np.array_equal(graph[key]["t"], t)
        assert sum(b.stage_seconds.values()) > 0
    finally:
        b.close()


def small_scene():
    rng = np.random.default_rng(43)
    views, poses, cam = S.make_feature_views(rng, n_views=8, n_points=1500, n_clutter=500, desc_noise=0.012)
    V = len(views)
    sim = np.zeros((V, V))
    pairs = []
    for i in range(V):
        for j in range(i + 1, V):
            shared = len(set(views[i]["point_id"][views[i]["point_id"] >= 0]) & set(views[j]["point_id"]))
            sim[i, j] = sim[j, i] = round(0.2 + 0.7 * shared / 1500 + 0.001 * ((3 * i + j) % 7), 3)
            pairs.append((i, j, sim[i, j]))
    # one view with few keypoints: its first pair is descriptor-matched and committed; with epipolar hashing its later pairs find
    # 30-51 tracklet matches -- quick pairs below kMinimumPointNumber, counted and skipped (:550-551), and one just above it
    views[V - 1] = {k: v[:130] for k, v in views[V - 1].items()}
    return views, poses, cam, sim, pairs


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["1", "2", "3", "4"])
def test_feature_pipeline_equals_the_cpu_restatement(tmp_path, mode):
    """The whole feature-level run -- quick matching from tracklets, descriptor matching, createCorrespondenceMatrix, A*
    guesses and their screening, estimatePose, commits, guided matching on the new poses, tracklet updates -- against
    oracle/pipeline_oracle.py, a CPU-only restatement over the other oracles: every counter equal, the same edges with the same
    scores, poses to 1e-9 (the decomposition's last bits differ between device and oracle by <= 1e-13 and chained guesses
    inherit them).  Modes: 1 path finding, 2 + epipolar hashing (tracklets in HBM), 3 the same with the host store, 4 with
    rotation-guided re-estimation of the chained poses."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_lib as O
    import pipeline_oracle as PO
    views, poses, cam, sim, pairs = small_scene()
    wave = 4
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    SC.write_feature_scene(fin, views, cam, sim, pairs, wave)
    r = subprocess.run([EXE, fin, fout, mode], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    (st, edges), = parse(open(fout, "rb").read(), 1)
    got = dict(zip(SC.PIPELINE_KEYS, st))
    lookup = lambda p, q: 1.0 if p == q else float(sim[p, q])
    ref, ref_edges, fragile = PO.run_features(O, views, cam, pairs, lookup, wave, path_finding=True, hashing=mode != "1",
                                              rotation_guided=mode == "4")
    assert fragile == 0   # (no guided-matching row sat in the 1e-7 don't-care band of a bin edge; otherwise pick another seed)
    for key in SC.PIPELINE_KEYS:
        assert got[key] == ref[key], (key, got[key], ref[key])
    assert got["waves"] >= 6 and got["paths_found"] > 0 and got["poses_from_guess"] > 0
    if mode != "1":   # skipped quick pairs; guided matching cut at kMaximumPointNumberForEpipolarHashing on all pairs but one
        assert got["too_few_matches"] >= 4 and got["quick_matching_runs"] > got["guided_matching_runs"] > 0 and got["track_number"] > 0
        assert 100 * (got["guided_matching_runs"] - 1) <= got["guided_matches_added"] < 100 * got["guided_matching_runs"]
    assert edges.keys() == ref_edges.keys() and len(edges) >= 21
    for key, (sc, R, t) in edges.items():
        rs, rR, rt = ref_edges[key]
        assert sc == rs, key
        assert np.abs(R.reshape(3, 3) - rR).max() < 1e-9 and np.abs(t - rt).max() < 1e-9, key
