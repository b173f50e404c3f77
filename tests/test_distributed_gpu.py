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


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def pair_similarity(g):
    """What the retrieval network would give: higher for pairs that share more scene (tests/test_scheduler.py)."""
    b = g["batch"]
    out = np.zeros(len(g["pairs"]))
    for e, (i, j) in enumerate(g["pairs"]):
        a, z = int(b["offsets"][e]), int(b["offsets"][e + 1])
        out[e] = round(0.3 + 0.6 * b["inlier"][a:z].mean() + 0.05 * ((int(i) * 7 + int(j)) % 3), 3)
    return out


def write_scene(path, g, wave, sim_kind):
    """u32 V, P, wave, simKind | [V x V f64 similarity if simKind == 1] | per pair: u32 src, dst, n; f64 thr, similarity;
    n x 4 f64 rows (cv::Mat N x 4 CV_64F, the reference's correspondence matrix)."""
    b, V = g["batch"], len(g["R_gt"])
    sim = pair_similarity(g)
    with open(path, "wb") as f:
        f.write(struct.pack("<IIII", V, len(g["pairs"]), wave, sim_kind))
        if sim_kind == 1:
            dense = np.zeros((V, V))
            for e, (i, j) in enumerate(g["pairs"]):
                dense[i, j] = dense[j, i] = sim[e]
            f.write(dense.astype("<f8").tobytes())
        for e, (i, j) in enumerate(g["pairs"]):
            a, z = int(b["offsets"][e]), int(b["offsets"][e + 1])
            f.write(struct.pack("<IIIdd", int(i), int(j), z - a, 7.5e-4, sim[e]))
            f.write(np.stack([b["x1"][a:z], b["y1"][a:z], b["x2"][a:z], b["y2"][a:z]], 1).astype("<f8").tobytes())


def run_ranks(cmd, world, timeout=1500):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    for rc, o, e in outs:
        assert rc == 0, (rc, o[-2000:], e[-2000:])
    return [o for _, o, _ in outs]


SCENES = {
    # name: (V, k, make_scene_graph overrides, wave size)
    "v340": (340, 12, dict(median_corr=500, max_corr=3000), 512),
    "v5000": (5000, 4, dict(median_corr=100, min_corr=60, max_corr=400, ring=3), 4096),  # ring edges keep it connected
}


@pytest.fixture(scope="module", params=list(SCENES))
def scene(request, tmp_path_factory):
    V, k, kw, wave = SCENES[request.param]
    g = S.make_scene_graph(V, k=k, seed=11, outlier_pair_frac=0.03, **kw)
    d = tmp_path_factory.mktemp(request.param)
    path = str(d / "scene.bin")
    write_scene(path, g, wave, sim_kind=2)
    return dict(name=request.param, g=g, path=path, dir=d, V=V, wave=wave)


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
    assert RO.align_error_deg(_rotations(single, V), g["R_gt"]).mean() < (0.5 if V < 1000 else 1.5)
    # uneven blocks really happened (row-balanced cut of ragged pairs)
    from pyposegraphbuilder import distributed as D
    lo_hi = D.shard_bounds(np.diff(g["batch"]["offsets"].astype(np.int64)), 2)
    assert lo_hi[0][1] - lo_hi[0][0] != lo_hi[1][1] - lo_hi[1][0]


@pytest.mark.gpu
def test_config5_wave_protocol_with_astar(scene):
    import rotavg_oracle as RO
    g, V, d = scene["g"], scene["V"], scene["dir"]
    run_ranks([EXE, scene["path"], str(d / "waves_w1"), "waves"], 1)
    run_ranks([EXE, scene["path"], str(d / "waves_w2"), "waves"], 2)
    single = open(str(d / "waves_w1.0"), "rb").read()
    assert open(str(d / "waves_w2.0"), "rb").read() == single and open(str(d / "waves_w2.1"), "rb").read() == single
    st = struct.unpack_from("<12Q", single, 0)
    # pairs processed, edges added == graph edges, A* searched / found / touched, poses from guesses, waves
    assert st[0] == len(g["pairs"]) and st[1] == st[8] and st[7] >= 2
    assert st[2] > 0 and st[3] > 0 and st[5] > 0 and st[11] == st[4]   # RunningStatistics "[A*] Touched nodes" agrees
    assert st[10] == st[8] and st[9] > 0
    err = RO.align_error_deg(_rotations(single, V), g["R_gt"])
    print("config 5 %s: %d edges, %d from A* guesses, global rotation error mean %.3f median %.3f deg" % (
        scene["name"], st[8], st[5], err.mean(), np.median(err)))
    # With the reference's un-squared getInliers bound (graph_traversal.h:164, guess_quirk = 1) a chained pose is accepted
    # on a wrongly retrieved pair too (random rows fall inside the ~24 px band) and the edge carries a high score; the
    # densely connected V = 340 graph averages that away, the thin V = 5000 ring does not (DESIGN.md, quirk ledger).
    assert err.mean() < 0.5 if V < 1000 else np.median(err) < 45.0


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
    st = struct.unpack_from("<12Q", single, 0)
    err = RO.align_error_deg(_rotations(single, V), g["R_gt"])
    print("config 5 guided %s: %d edges, %d from rotation-guided guesses of %d searched, %d hypotheses, rotation error mean %.3f deg" % (
        scene["name"], st[8], st[5], st[2], st[6], err.mean()))
    assert st[5] > 0.5 * st[3] > 0                     # most chained rotations lead to an accepted edge
    assert err.mean() < (0.5 if V < 1000 else 1.5)
    if os.path.exists(str(d / "waves_w1.0")):          # fewer hypotheses than the reference-style run of the same scene
        ref = struct.unpack_from("<12Q", open(str(d / "waves_w1.0"), "rb").read(), 0)
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
