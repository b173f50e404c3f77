"""BASELINE configs 4 and 5 with more than one rank, on the one GPU the box has.

Two freshly spawned processes share the visible GPU (RCCL refuses duplicate devices, so the edge records travel over the
host transport of pgi_allgather_edges: the C++ TCP star, or gloo from Python) and must reproduce the single-process
result BIT FOR BIT: the gathered edge table / pose graph, the scheduler statistics and the global rotations.
  config 4: pairs sharded (uneven, row-balanced blocks) -> estimate -> all-gather -> replicated rotation averaging
  config 5: A*-scheduled waves; every wave is sharded, host A* runs on the committed snapshot, records are gathered,
            every rank commits the whole wave
Scene graphs: V = 340 (Madrid-Metropolis-sized surrogate) and V = 5000 (Trafalgar-sized surrogate); 1DSfM data is not
available on either box."""
import os
import socket
import struct
import subprocess
import sys

import numpy as np
import pytest

from pyposegraphbuilder import synthetic as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
EXE = os.path.join(ROOT, "pose-graph-initialization_amd", "test_distributed")
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")


import scene_drivers as SC
from scene_drivers import write_scene, run_ranks, SCENES  # noqa: E402  (scene format, launcher: shared with bench.py)


# v340 / v5000: SURVEY 8d's density (k ~ 40 nearest views, median ~ 600 rows per pair, cap 8000: ~7 300 / ~106 000 pairs,
# ~5 M / ~77 M rows); v5000_ring: the thin graph of rounds 2-3 (~3 edges per view), where the reference's guess quirk shows
@pytest.fixture(scope="module", params=["v340", "v5000", "v5000_ring"])
def scene(request, tmp_path_factory):
    V, k, kw, wave, dense = SCENES[request.param]
    g, wave = SC.make_scene(request.param)
    d = tmp_path_factory.mktemp(request.param)
    path = str(d / "scene.bin")
    (SC.write_scene_bulk if dense else write_scene)(path, g, wave, sim_kind=2)
    return dict(name=request.param, g=g, path=path, dir=d, V=V, wave=wave, dense=dense)


CONFIG5_REFERENCE_GUESS_BOUND_DEG = {"v340": 0.5, "v5000": 2.0}   # mean global rotation error with the reference's guess path (measured 0.11 / 0.74)


def _rotations(blob, V):
    return np.frombuffer(blob[-V * 72:], "<f8").reshape(V, 3, 3)


@pytest.mark.gpu
def test_config4_sharded_estimate_gather_average(scene):
    import rotavg_oracle as RO
    g, V, d = scene["g"], scene["V"], scene["dir"]
    P = len(g["pairs"])
    run_ranks([EXE, scene["path"], str(d / "shard_w1"), "shard"], 1)
    o2 = run_ranks([EXE, scene["path"], str(d / "shard_w2"), "shard"], 2)
    assert all("transport host" in o for o in o2)           # one GPU: the records go through the host transport
    single = open(str(d / "shard_w1.0"), "rb").read()
    r0, r1 = open(str(d / "shard_w2.0"), "rb").read(), open(str(d / "shard_w2.1"), "rb").read()
    assert r0 == single and r1 == single                    # edge table, iteration counts and rotations: bit for bit
    hdr = struct.unpack_from("<4Q", single, 0)
    assert hdr[0] == P and hdr[1] == hdr[3] and hdr[1] >= 0.9 * int((~g["wrong"]).sum()) and hdr[2] > 0
    from pyposegraphbuilder._lib import EDGE_DTYPE
    edges = np.frombuffer(single, EDGE_DTYPE, P, 32)
    ok = edges["status"] == 1
    err = np.array([S.rot_err_deg(edges["R"][e].reshape(3, 3), g["batch"]["R"][e]) for e in np.nonzero(ok)[0]])
    assert np.mean(err < 1.0) > 0.8
    # the averaged rotations are right, too (errors accumulate along the 5000-view ring: looser bound there)
    gerr = RO.align_error_deg(_rotations(single, V), g["R_gt"])
    print("config 4 %s: %d pairs, %d rows, %d edges, AUC@5 %.4f, global rotation error mean %.4f deg" % (
        scene["name"], P, int(g["batch"]["offsets"][-1]), hdr[1],
        S.auc_at(np.where(ok, np.array([S.rot_err_deg(edges["R"][e].reshape(3, 3), g["batch"]["R"][e]) for e in range(P)]), np.inf)[~g["wrong"]], 5.0),
        gerr.mean()))
    assert gerr.mean() < (1.5 if scene["name"] == "v5000_ring" else 0.5)
    if V <= 340:
        # ... and they are the oracle's for these edges (sparse direct solves; weight = inliers / rows as estimateAndAverage sets it)
        rows = np.diff(g["batch"]["offsets"].astype(np.int64))
        Ro, iters_o = RO.rotation_average(V, g["pairs"][ok, 0], g["pairs"][ok, 1], edges["R"][ok].reshape(-1, 3, 3),
                                          edges["n_inl"][ok] / np.maximum(rows[ok], 1))
        dR = np.einsum("kij,kmj->kim", _rotations(single, V), Ro)
        ang = np.arccos(np.clip((np.trace(dR, axis1=1, axis2=2) - 1) / 2, -1, 1))
        assert ang.max() < 1e-5 and abs(int(hdr[2]) - iters_o) <= 1, (ang.max(), hdr[2], iters_o)
    # uneven blocks really happened (row-balanced cut of ragged pairs)
    from pyposegraphbuilder import distributed as D
    lo_hi = D.shard_bounds(np.diff(g["batch"]["offsets"].astype(np.int64)), 2)
    assert lo_hi[0][1] - lo_hi[0][0] != lo_hi[1][1] - lo_hi[1][0]


@pytest.mark.gpu
def test_dense_scene_sample_equals_the_oracle(scene):
    """K1 on the scene's own rows (ragged: 60 ... 8000 rows per pair, every occupancy class, wrongly retrieved pairs among
    them): a sample of >= 10 000 pairs of the V = 5000 scene (every pair of the smaller ones, capped), masks / E / counts
    bit-identical to the CPU oracle."""
    if not scene["dense"]:
        pytest.skip("sampled on the dense scenes")
    import oracle_lib as O
    from pyposegraphbuilder import Engine
    g = scene["g"]
    P = len(g["pairs"])
    idx = np.arange(0, P, max(1, P // 10500))[:10500] if P > 12000 else np.arange(min(P, 4000))
    b = S.take_pairs(g, idx)
    eng = Engine()
    try:
        db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, seed=7)
        edges, masks = eng.estimate_pose_batch(db)
        got = eng.edges_to_numpy(edges)
        gm = masks.cpu().numpy()
    finally:
        eng.close()
    exp, emasks = O.estimate_pose_batch(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, O.default_params(), 7)
    assert len(idx) >= (10000 if P > 12000 else 1)
    assert np.array_equal(gm, emasks)
    for key in ("E", "status", "n_inl", "iters", "lo_runs", "score"):
        assert np.array_equal(got[key], exp[key]), key
    ok = got["status"] == 1
    for i in np.nonzero(ok)[0][:2000]:
        assert S.rot_err_deg(got["R"][i].reshape(3, 3), exp["R"][i].reshape(3, 3)) < 1e-4 * 57.3


@pytest.mark.gpu
def test_config5_wave_protocol_with_astar(scene):
    import rotavg_oracle as RO
    g, V, d = scene["g"], scene["V"], scene["dir"]
    run_ranks([EXE, scene["path"], str(d / "waves_w1"), "waves"], 1)
    run_ranks([EXE, scene["path"], str(d / "waves_w2"), "waves"], 2)
    single = open(str(d / "waves_w1.0"), "rb").read()
    assert open(str(d / "waves_w2.0"), "rb").read() == single and open(str(d / "waves_w2.1"), "rb").read() == single
    st = struct.unpack_from("<13Q", single, 0)
    # pairs processed, edges added == graph edges, A* searched / found / touched, poses from guesses, waves
    assert st[0] == len(g["pairs"]) and st[1] == st[8] and st[7] >= 2
    assert st[2] > 0 and st[3] > 0 and st[5] > 0 and st[11] == st[4]   # RunningStatistics "[A*] Touched nodes" agrees
    assert st[10] == st[8] and st[9] > 0
    err = RO.align_error_deg(_rotations(single, V), g["R_gt"])
    stats, graph_edges = SC.read_waves(single)
    # edges whose rotation is wrong by more than 5 degrees, and how many of them came in through the quirk alone
    lut = {(int(i), int(j)): e for e, (i, j) in enumerate(g["pairs"])}
    bad = sum(1 for ge in graph_edges
              if S.rot_err_deg(ge["R"].reshape(3, 3), g["batch"]["R"][lut[(int(ge["src"]), int(ge["dst"]))]]) > 5.0)
    print("config 5 %s: %d edges, %d from A* guesses (%d of them only through the un-squared bound), %d edges off by > 5 deg, "
          "global rotation error mean %.3f median %.3f deg" % (scene["name"], st[8], st[5], st[12], bad, err.mean(), np.median(err)))
    # With the reference's un-squared getInliers bound (graph_traversal.h:164, guess_quirk = 1) a chained pose is accepted
    # on a wrongly retrieved pair too (random rows fall inside the ~24 px band) and the edge carries a high score; the
    # densely connected V = 340 graph averages that away, the thin V = 5000 ring does not (DESIGN.md, quirk ledger).
    # The run now COUNTS those edges ("[Pose estimation] Quirk-only guesses"): accepted guesses whose inlier count under the
    # squared bound (1.5 thr)^2 is below kMinimumInlierNumber.
    assert st[12] <= st[5]
    if scene["dense"]:
        # ~40 edges per view: the averaging outvotes the wrongly accepted chained poses (they are still counted)
        assert err.mean() < CONFIG5_REFERENCE_GUESS_BOUND_DEG[scene["name"]]
    else:
        assert np.median(err) < 45.0
        # the count explains the damage: the wrong edges of the graph are (nearly all) quirk-only guesses -- a wrongly
        # retrieved pair has no other way in -- and there are enough of them to break the thin ring
        assert st[12] > 0 and bad > 0 and st[12] >= 0.8 * bad


@pytest.mark.gpu
def test_config5_rotation_guided_reestimation(scene):
    """BASELINE config 5 as named: A*-scheduled edges + rotation-guided re-estimation (guess_mode 1): the chain's rotation
    is kept, the translation direction re-estimated from two-point hypotheses.  Two ranks == one rank bit for bit, and --
    unlike the reference's guess path -- the thin V = 5000 ring comes out right, with far fewer hypotheses drawn."""
    import rotavg_oracle as RO
    g, V, d = scene["g"], scene["V"], scene["dir"]
    run_ranks([EXE, scene["path"], str(d / "guided_w1"), "waves_guided"], 1)
    run_ranks([EXE, scene["path"], str(d / "guided_w2"), "waves_guided"], 2)
    single = open(str(d / "guided_w1.0"), "rb").read()
    assert open(str(d / "guided_w2.0"), "rb").read() == single and open(str(d / "guided_w2.1"), "rb").read() == single
    st = struct.unpack_from("<13Q", single, 0)
    err = RO.align_error_deg(_rotations(single, V), g["R_gt"])
    assert st[12] == 0                                 # no reference-style guess screening in this mode: nothing to count
    print("config 5 guided %s: %d edges, %d from rotation-guided guesses of %d searched, %d hypotheses, rotation error mean %.3f deg" % (
        scene["name"], st[8], st[5], st[2], st[6], err.mean()))
    assert st[5] > 0.5 * st[3] > 0                     # most chained rotations lead to an accepted edge
    assert err.mean() < (1.0 if scene["name"] == "v5000_ring" else 0.5)     # (r02 measured 0.72 deg on the thin ring)
    if os.path.exists(str(d / "waves_w1.0")):          # fewer hypotheses than the reference-style run of the same scene
        ref = struct.unpack_from("<13Q", open(str(d / "waves_w1.0"), "rb").read(), 0)
        assert st[6] < ref[6]


@pytest.mark.gpu
def test_python_ranks_over_gloo_match_single_process(tmp_path):
    """pyposegraphbuilder.distributed.Communicator (gloo bootstrap, host transport of pgi_allgather_edges): shard ->
    estimate -> gather -> pgi_rotation_average_edges from the device table, world 2 == world 1, bit for bit."""
    out1, out2 = str(tmp_path / "w1"), str(tmp_path / "w2")
    run_ranks([sys.executable, WORKER, out1], 1)
    o = run_ranks([sys.executable, WORKER, out2], 2)
    assert all("transport=host" in x for x in o)
    single = open(out1 + ".0", "rb").read()
    assert open(out2 + ".0", "rb").read() == single and open(out2 + ".1", "rb").read() == single


@pytest.mark.gpu
def test_rccl_two_ranks_match_single_process(tmp_path):
    """The RCCL transport with MORE than one rank (ncclAllGather is never reached here -- the row-balanced blocks are
    uneven -- so this is the grouped ncclSend/ncclRecv path of csrc/pgi_comm.hip): two ranks on two devices reproduce
    the single-process table and rotations bit for bit.  Needs two GPUs; the one-GPU box skips (RCCL refuses ranks
    that share a device) and covers the same protocol over the host transport in the test above."""
    from pyposegraphbuilder import _lib as L
    n_dev = L.load().pgi_device_count()
    if n_dev < 2:
        pytest.skip("pgi_device_count() = %d: RCCL needs one device per rank" % n_dev)
    out1, out2 = str(tmp_path / "w1"), str(tmp_path / "w2")
    run_ranks([sys.executable, WORKER, out1], 1)
    o = run_ranks([sys.executable, WORKER, out2], 2, extra_env={"PGI_TEST_RCCL": "1"})
    assert all("transport=rccl" in x for x in o)
    single = open(out1 + ".0", "rb").read()
    assert open(out2 + ".0", "rb").read() == single and open(out2 + ".1", "rb").read() == single


@pytest.mark.gpu
def test_python_builder_run_is_the_cpp_scheduler(tmp_path):
    """pyposegraphbuilder.PoseGraphBuilder.run goes through libpgi_host.so (pgih_run_pairs, include/pgi_host.h): on the
    V = 340 scene it returns the graph the C++ driver's A*-scheduled run writes -- same edges, scores, rotations,
    translations, bit for bit -- and the same scheduler counters; with rotationGuided the guided driver's."""
    from pyposegraphbuilder import PoseGraphBuilder
    g, wave = SC.make_scene("v340_thin")
    path = str(tmp_path / "scene.bin")
    write_scene(path, g, wave, sim_kind=2)
    b, sim = g["batch"], SC.pair_similarity(g)
    pairs = []
    for e, (i, j) in enumerate(g["pairs"]):
        a, z = int(b["offsets"][e]), int(b["offsets"][e + 1])
        pairs.append(dict(src=int(i), dst=int(j), similarity=float(sim[e]), threshold=7.5e-4,
                          correspondences=np.stack([b["x1"][a:z], b["y1"][a:z], b["x2"][a:z], b["y2"][a:z]], 1)))
    builder = PoseGraphBuilder(20, 5000, 5, 100, 20, 50, 100, 0.8, 0.05, 0.4, "", "", "", "", True, True, True)
    try:
        for mode, guided in (("waves", False), ("waves_guided", True)):
            run_ranks([EXE, path, str(tmp_path / mode), mode], 1)
            stats, edges = SC.read_waves(open(str(tmp_path / mode) + ".0", "rb").read())
            graph = builder.run(pairs, waveSize=wave, rotationGuided=guided, numViews=len(g["R_gt"]))
            assert len(graph) == len(edges) == stats["graph_edges"]
            for r in edges:
                ge = graph[(int(r["src"]), int(r["dst"]))]
                assert ge["score"] == r["score"] and np.array_equal(ge["R"].ravel(), r["R"]) and np.array_equal(ge["t"], r["t"])
            for key in ("pairs_processed", "edges_added", "paths_searched", "paths_found", "touched_nodes", "poses_from_guess",
                        "hypotheses", "waves", "quirk_only_guesses"):
                assert builder.statistics[key] == stats[key], key
            assert builder.statistics["paths_found"] > 0 and builder.statistics["poses_from_guess"] > 0
    finally:
        builder.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["shard", "waves_guided"])
def test_eight_ranks_over_the_host_transport_match_single_process(scene, mode):
    """The width BASELINE's configs 4/5 name: EIGHT ranks (processes) of the C++ driver sharing the one GPU of the box, edge
    records over the host transport (RCCL refuses ranks that share a device): the shard / gather / commit protocol at world 8
    reproduces the single-process result byte for byte on every rank.  (Round 4 never ran it wider than 3.)"""
    if scene["name"] != "v340":
        pytest.skip("eight processes on one GPU: the V = 340 scene")
    d = scene["dir"]
    run_ranks([EXE, scene["path"], str(d / ("w1_" + mode)), mode], 1)
    o8 = run_ranks([EXE, scene["path"], str(d / ("w8_" + mode)), mode], 8)
    assert len(o8) == 8 and all("transport host" in o and "/8 " in o for o in o8)
    single = open(str(d / ("w1_" + mode)) + ".0", "rb").read()
    for r in range(8):
        assert open(str(d / ("w8_" + mode)) + ".%d" % r, "rb").read() == single, r
    # eight uneven, row-balanced, non-empty blocks really happened
    from pyposegraphbuilder import distributed as D
    lo_hi = D.shard_bounds(np.diff(scene["g"]["batch"]["offsets"].astype(np.int64)), 8)
    assert len({hi - lo for lo, hi in lo_hi}) > 1 and all(hi > lo for lo, hi in lo_hi)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["waves", "waves_guided"])
def test_tail_wave_with_fewer_pairs_than_ranks(tmp_path, mode):
    """the scheduler's next-wave prefetch on a rank whose block of the NEXT wave is empty.  Wave size =
    all candidates but two, four ranks: the tail wave has two pairs, so at least two ranks own nothing of it while they prefetch
    it from inside the first wave's call.  Such a prefetch must upload nothing and touch no collective: every rank's result
    equals the single-process run's byte for byte (before the fix the helper thread ran the full estimate path: a second
    all-gather the peers never issued)."""
    from pyposegraphbuilder import distributed as D
    g, _wave = SC.make_scene("v340_thin")
    P = len(g["pairs"])
    wave = P - 2
    path = str(tmp_path / "scene.bin")
    write_scene(path, g, wave, sim_kind=2)
    run_ranks([EXE, path, str(tmp_path / "w1"), mode], 1)
    single = open(str(tmp_path / "w1.0"), "rb").read()
    st = struct.unpack_from("<13Q", single, 0)
    assert st[0] == P and st[7] >= 2, st                      # every pair processed, in at least two waves
    sizes = np.diff(g["batch"]["offsets"].astype(np.int64))
    o4 = run_ranks([EXE, path, str(tmp_path / "w4"), mode], 4)
    assert len(o4) == 4
    for r in range(4):
        assert open(str(tmp_path / ("w4.%d" % r)), "rb").read() == single, r
    # (the premise: a wave of two pairs leaves at least two of four blocks empty)
    assert sum(1 for lo, hi in D.shard_bounds(sizes[-2:], 4) if hi == lo) >= 2


@pytest.mark.gpu
@pytest.mark.parametrize("guided", [False, True])
def test_c_abi_run_pairs_is_the_driver_on_the_dense_scene(scene, guided):
    """pgih_run_pairs of libpgi_host.so (include/pgi_host.h) -- the installed entry point a non-C++ caller binds -- on the
    DENSE V = 5000 scene (106 151 pairs, 76.6 M rows handed over as ONE flat N x 4 CV_64F block): the pose graph it returns is
    the one tests/cpp/test_distributed.cpp's scheduled run writes, edge for edge in insertion order, bit for bit, with the same
    scheduler counters.  (Round 4 compared the two on the thin V = 340 scene only; bench.py's graph legs time the driver.)"""
    import ctypes as C
    from pyposegraphbuilder import PoseGraphBuilder
    if scene["name"] != "v5000":
        pytest.skip("the dense V = 5000 scene")
    g, d, V, wave = scene["g"], scene["dir"], scene["V"], scene["wave"]
    mode = "waves_guided" if guided else "waves"
    out = str(d / ("abi_" + mode))
    run_ranks([EXE, scene["path"], out, mode], 1)
    stats, edges = SC.read_waves(open(out + ".0", "rb").read())
    b = g["batch"]
    P = len(g["pairs"])
    corr = np.empty((int(b["offsets"][-1]), 4), np.float64)
    for c, k in enumerate(("x1", "y1", "x2", "y2")):
        corr[:, c] = b[k]
    src, dst = np.ascontiguousarray(g["pairs"][:, 0], np.uint32), np.ascontiguousarray(g["pairs"][:, 1], np.uint32)
    sim, thr = np.ascontiguousarray(SC.pair_similarity(g), np.float64), np.full(P, 7.5e-4)
    off = np.ascontiguousarray(b["offsets"], np.uint64)
    rec = np.zeros(P, [("src", "<u4"), ("dst", "<u4"), ("score", "<f8"), ("R", "<f8", 9), ("t", "<f8", 3)])
    n_edges, st = C.c_uint32(0), np.zeros(16, np.uint64)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    builder = PoseGraphBuilder(20, 5000, 5, 100, 20, 50, 100, 0.8, 0.05, 0.4, "", "", "", "", True, True, True)
    try:
        lib, h = builder._host()
        assert lib.pgih_set_rotation_guided(h, int(guided)) >= 0
        rc = lib.pgih_run_pairs(h, V, P, ptr(src), ptr(dst), ptr(sim), ptr(thr), ptr(off), ptr(corr), wave, 0, ptr(rec), P,
                                C.byref(n_edges), ptr(st))
        assert rc >= 0, lib.pgih_last_error()
    finally:
        builder.close()
    assert n_edges.value == len(edges) == stats["graph_edges"] > 0.9 * int((~g["wrong"]).sum())
    got = rec[:n_edges.value]
    for key in ("src", "dst", "score", "R", "t"):
        assert np.array_equal(got[key], edges[key]), key
    for k, key in enumerate(("pairs_processed", "edges_added", "paths_searched", "paths_found", "touched_nodes", "poses_from_guess",
                             "hypotheses", "waves", "graph_edges", "quirk_only_guesses")):
        want = stats["quirk_only_guesses"] if key == "quirk_only_guesses" else stats[key]
        assert int(st[k]) == want, (key, int(st[k]), want)
