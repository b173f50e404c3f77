"""Wave scheduler with A* pose guesses (BASELINE config 5 surrogate) on the GPU."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from pyposegraphbuilder import synthetic as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import rotavg_oracle as RO  # noqa: E402

EXE = os.path.join(ROOT, "pose-graph-initialization_amd", "test_scheduler")


@pytest.mark.gpu
def test_wave_scheduler_with_astar_guesses(tmp_path):
    V = 80
    g = S.make_scene_graph(V, k=8, seed=5, outlier_pair_frac=0.03)
    b = g["batch"]
    # image similarity: high for neighbouring views on the ring (what the retrieval network would give)
    sim = np.zeros((V, V))
    for e, (i, j) in enumerate(g["pairs"]):
        a, z = int(b["offsets"][e]), int(b["offsets"][e + 1])
        sim[i, j] = sim[j, i] = round(0.3 + 0.6 * b["inlier"][a:z].mean() + 0.05 * ((i * 7 + j) % 3), 3)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<III", V, len(g["pairs"]), 64))
        f.write(sim.astype("<f8").tobytes())
        for e, (i, j) in enumerate(g["pairs"]):
            a, z = int(b["offsets"][e]), int(b["offsets"][e + 1])
            f.write(struct.pack("<IIIdd", i, j, z - a, 7.5e-4, sim[i, j]))
            f.write(np.stack([b["x1"][a:z], b["y1"][a:z], b["x2"][a:z], b["y2"][a:z]], 1).astype("<f8").tobytes())
    r = subprocess.run([EXE, fin, fout], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    buf = open(fout, "rb").read()
    pos = 0
    res = []
    for _ in range(2):
        st = struct.unpack_from("<9Q", buf, pos)
        pos += 72
        edges = []
        for _e in range(st[8]):
            s, d, sc = struct.unpack_from("<IId", buf, pos)
            R = np.frombuffer(buf, "<f8", 9, pos + 16).reshape(3, 3)
            pos += 88
            edges.append((s, d, sc, R))
        res.append((st, edges))
    assert pos == len(buf)
    (st0, e0), (st1, e1) = res
    n_good = int((~g["wrong"]).sum())
    for st, edges in res:
        assert st[0] == len(g["pairs"]) and st[1] == st[8] and st[1] >= 0.9 * n_good
        truth = {(int(i), int(j)): b["R"][e] for e, (i, j) in enumerate(g["pairs"])}
        err = np.array([S.rot_err_deg(R, truth[(s, d)]) for s, d, sc, R in edges])
        assert np.mean(err < 1.0) > 0.8  # wide-baseline, low-inlier and wrongly retrieved pairs are in the mix
        # pose graph -> global rotations
        src = np.array([x[0] for x in edges]); dst = np.array([x[1] for x in edges])
        Rg, _ = RO.rotation_average(V, src, dst, np.stack([x[3] for x in edges]), np.array([x[2] for x in edges]))
        assert RO.align_error_deg(Rg, g["R_gt"]).mean() < 0.5
    # without path finding: no searches, no guesses
    assert st0[2] == 0 and st0[5] == 0
    # with path finding: later waves search the committed graph, chained poses are accepted and
    # replace the robust fit (fewer hypotheses drawn overall)
    assert st1[2] > 0 and st1[3] > 0 and st1[5] > 0.3 * st1[3]
    assert st1[6] < 0.8 * st0[6]
    print("no-A*: %d hyps; A*: searched %d found %d used %d, %d hyps" % (st0[6], st1[2], st1[3], st1[5], st1[6]))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["waves", "waves_guided"])
def test_wave_scheduler_equals_the_cpu_restatement(tmp_path, mode):
    """The whole scheduled run -- candidate order, waves, visibility, A* guesses on the committed graph, the 5-inlier
    screening, estimatePose with and without guesses, commits -- against oracle/scheduler_oracle.py, a CPU-only restatement
    built from the other oracles (A*, K2's scoring, pgo_estimate_pose_batch): every scheduler counter equal, the same edges in
    the same order with the same scores, rotations and translations to 1e-9 (the decomposition's last bits differ between
    device and oracle by <= 1e-13, which the chained guesses inherit)."""
    import oracle_lib as O
    import scheduler_oracle as SO
    import scene_drivers as SC
    V, wave = 80, 64
    g = S.make_scene_graph(V, k=8, seed=5, outlier_pair_frac=0.03)
    path = str(tmp_path / "scene.bin")
    SC.write_scene(path, g, wave, sim_kind=2)
    SC.run_ranks([SC.EXE, path, str(tmp_path / mode), mode], 1, extra_env={"PGI_QUIET": "1"})
    stats, edges = SC.read_waves(open(str(tmp_path / mode) + ".0", "rb").read())
    b, sim = g["batch"], SC.pair_similarity(g)
    table = {}
    pairs = []
    for e, (i, j) in enumerate(g["pairs"]):
        a, z = int(b["offsets"][e]), int(b["offsets"][e + 1])
        table[(int(i), int(j))] = table[(int(j), int(i))] = float(sim[e])
        pairs.append(dict(src=int(i), dst=int(j), similarity=float(sim[e]), thr=7.5e-4, x1=b["x1"][a:z], y1=b["y1"][a:z],
                          x2=b["x2"][a:z], y2=b["y2"][a:z]))
    lookup = lambda p, q: 1.0 if p == q else table.get((p, q), 0.0)
    st, ref = SO.run_waves(O, pairs, lookup, V, wave, path_finding=True, rotation_guided=(mode == "waves_guided"))
    for key in ("pairs_processed", "edges_added", "paths_searched", "paths_found", "touched_nodes", "poses_from_guess", "hypotheses",
                "waves", "graph_edges", "quirk_only_guesses"):
        assert stats[key] == st[key], (key, stats[key], st[key])
    assert stats["paths_found"] > 0 and stats["poses_from_guess"] > 0 and stats["waves"] >= 3
    assert len(edges) == len(ref)
    for got, (s_, d_, score, R, t) in zip(edges, ref):
        assert (int(got["src"]), int(got["dst"])) == (s_, d_) and got["score"] == score
        assert np.abs(got["R"].reshape(3, 3) - R).max() < 1e-9 and np.abs(got["t"] - t).max() < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["waves", "waves_guided"])
def test_next_wave_prefetch_equals_the_sequential_scheduler(tmp_path, mode):
    """Round 5: while a wave's kernels run, the scheduler forms the NEXT wave and uploads its rows into a second staging block
    (PoseGraphBuilder::run / estimatePoses `nextWave`).  The early formation sees the graph BEFORE the running wave's commit,
    which is wrong exactly for a candidate the list names twice (or once per direction) with the first instance in the
    running wave; the run detects that after the commit and forms the wave again.  Both cases against PGI_WAVE_PREFETCH=0,
    byte for byte: a plain scene (every wave prefetched) and the same scene with 24 candidates listed a second time --
    half of them reversed -- at similarities that put them one or two waves later."""
    import scene_drivers as SC
    V, wave = 80, 48
    g = S.make_scene_graph(V, k=8, seed=9, outlier_pair_frac=0.03)
    sim = SC.pair_similarity(g)
    b = g["batch"]
    off = b["offsets"].astype(np.int64)
    order = np.argsort(-sim, kind="stable")
    dup = order[np.arange(0, 24) * 3]                       # candidates of the first two waves
    pairs2 = g["pairs"][dup].copy()
    pairs2[::2] = pairs2[::2, ::-1]                         # every other one in the other direction
    rows = np.concatenate([np.arange(off[e], off[e + 1]) for e in dup])
    g2 = dict(g, pairs=np.concatenate([g["pairs"], pairs2]))
    g2["batch"] = {k: np.concatenate([b[k], b[k][rows]]) for k in ("x1", "y1", "x2", "y2", "inlier")}
    g2["batch"]["offsets"] = np.concatenate([off, off[-1] + np.cumsum(np.diff(off)[dup])]).astype(np.uint64)
    sim2 = np.concatenate([sim, np.round(sim[dup] - 0.004 * (1 + np.arange(24) % 3) - 0.1, 3)])
    for name, scene, sims in (("plain", g, None), ("twice", g2, sim2)):
        path = str(tmp_path / (name + ".bin"))
        SC.write_scene(path, scene, wave, sim_kind=2, sim=sims)
        SC.run_ranks([SC.EXE, path, str(tmp_path / (name + "_on")), mode], 1, extra_env={"PGI_QUIET": "1"})
        SC.run_ranks([SC.EXE, path, str(tmp_path / (name + "_off")), mode], 1, extra_env={"PGI_QUIET": "1", "PGI_WAVE_PREFETCH": "0"})
        on, offb = open(str(tmp_path / (name + "_on")) + ".0", "rb").read(), open(str(tmp_path / (name + "_off")) + ".0", "rb").read()
        assert on == offb, name
        stats, edges = SC.read_waves(on)
        assert stats["waves"] >= 4 and stats["paths_found"] > 0
        if name == "twice":   # the second listing of a pair that became an edge is skipped (pose_graph_builder.h:426-431)
            assert stats["pairs_processed"] < len(g2["pairs"]) and stats["pairs_processed"] >= len(g["pairs"]) - 2
            keys = {(int(e["src"]), int(e["dst"])) for e in edges}
            assert not any((int(d), int(s_)) in keys and (int(s_), int(d)) in keys for s_, d in g["pairs"][dup])


@pytest.mark.gpu
@pytest.mark.parametrize("guided", [False, True])
def test_python_builder_run_with_graph_cut_equals_the_cpu_restatement(guided):
    """pgih_set_graph_cut (include/pgi_host.h) through the Python builder: the A*-scheduled run with graph-cut local optimisation
    (lambda * 64 = 9) equals oracle/scheduler_oracle.py with the same switch -- same edges in the same order, scores equal,
    poses to 1e-9 -- differs from the default run, and the switch is restored afterwards."""
    import oracle_lib as O
    import scheduler_oracle as SO
    import scene_drivers as SC
    from pyposegraphbuilder import PoseGraphBuilder
    V, wave = 60, 48
    g = S.make_scene_graph(V, k=6, seed=9, outlier_pair_frac=0.03)
    b, sim = g["batch"], SC.pair_similarity(g)
    table, pairs_o, pairs_b = {}, [], []
    for e, (i, j) in enumerate(g["pairs"]):
        a, z = int(b["offsets"][e]), int(b["offsets"][e + 1])
        table[(int(i), int(j))] = table[(int(j), int(i))] = float(sim[e])
        pairs_o.append(dict(src=int(i), dst=int(j), similarity=float(sim[e]), thr=7.5e-4, x1=b["x1"][a:z], y1=b["y1"][a:z],
                            x2=b["x2"][a:z], y2=b["y2"][a:z]))
        pairs_b.append(dict(src=int(i), dst=int(j), similarity=float(sim[e]), threshold=7.5e-4,
                            correspondences=np.stack([b["x1"][a:z], b["y1"][a:z], b["x2"][a:z], b["y2"][a:z]], 1)))
    lookup = lambda p, q: 1.0 if p == q else table.get((p, q), 0.0)
    st, ref = SO.run_waves(O, pairs_o, lookup, V, wave, path_finding=True, rotation_guided=guided, graph_cut=9)
    builder = PoseGraphBuilder(20, 5000, 5, 100, 20, 50, 100, 0.8, 0.05, 0.4, "", "", "", "", True, True, True)
    try:
        graph = builder.run(pairs_b, waveSize=wave, rotationGuided=guided, numViews=V, graphCut=9)
        stats_gc = dict(builder.statistics)
        plain = builder.run(pairs_b, waveSize=wave, rotationGuided=guided, numViews=V)      # the switch did not stick
    finally:
        builder.close()
    assert len(graph) == len(ref) == stats_gc["graph_edges"]
    for key in ("pairs_processed", "edges_added", "paths_searched", "paths_found", "poses_from_guess", "hypotheses", "waves"):
        assert stats_gc[key] == st[key], (key, stats_gc[key], st[key])
    for (s_, d_, score, R, t), (key, e) in zip(ref, graph.items()):
        assert key == (s_, d_) and e["score"] == score
        assert np.abs(e["R"] - R).max() < 1e-9 and np.abs(e["t"] - t).max() < 1e-9
    st0, ref0 = SO.run_waves(O, pairs_o, lookup, V, wave, path_finding=True, rotation_guided=guided)
    assert len(plain) == len(ref0)
    for (s_, d_, score, R, t), (key, e) in zip(ref0, plain.items()):
        assert key == (s_, d_) and e["score"] == score and np.abs(e["R"] - R).max() < 1e-9
    assert any(not np.array_equal(graph[k]["R"], plain[k]["R"]) for k in graph if k in plain)   # the mode changes the refits
