"""TEST / MEASUREMENT TOOLING (not part of the product package; round 5 moved it out of pyposegraphbuilder/): scene-graph
surrogates of BASELINE configs 3/4/5 driven through the C++ host layer by the drivers built from tests/cpp/
(test_distributed.cpp, test_pipeline.cpp) -- the scene file format, the one-process-per-rank launcher and the readers of the
drivers' result files.  Shared by tests/test_distributed_gpu.py, bench.py (graph-level legs) and scripts/."""
import os
import socket
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pose-graph-initialization_amd")
import sys
if PKG not in sys.path:
    sys.path.insert(0, PKG)
from pyposegraphbuilder import synthetic as S  # noqa: E402
from pyposegraphbuilder._lib import EDGE_DTYPE  # noqa: E402
EXE = os.path.join(PKG, "test_distributed")

SCENES = {
    # name: (V, k, scene-graph overrides, wave size, dense generator?)
    # SURVEY 8d's density: candidate pairs = the k ~ 40 nearest views in view direction, N per pair ~ covisibility with median ~ 600, cap 8000
    "v340": (340, 40, dict(median_corr=600, max_corr=8000), 512, True),        # Madrid-Metropolis-sized: ~7 300 pairs, ~5.3 M rows
    "v5000": (5000, 40, dict(median_corr=600, max_corr=8000), 16384, True),    # Trafalgar-sized: ~106 000 pairs, ~77 M rows
    # the thin graphs of rounds 2-3 (k = 12 / k = 4 + ring 3, ~120 rows per pair at V = 5000): kept because a graph with ~3 edges
    # per view is where the reference's guess quirk breaks the averaged rotations (tests/test_distributed_gpu.py)
    "v340_thin": (340, 12, dict(median_corr=500, max_corr=3000), 512, False),
    "v5000_ring": (5000, 4, dict(median_corr=100, min_corr=60, max_corr=400, ring=3), 4096, False),
}


def make_scene(name, seed=11):
    V, k, kw, wave, dense = SCENES[name]
    if dense:
        return S.make_scene_graph_dense(V, k=k, seed=seed, outlier_pair_frac=0.03, **kw), wave
    return S.make_scene_graph(V, k=k, seed=seed, outlier_pair_frac=0.03, **kw), wave


def free_port():
    """A rendezvous port BELOW the kernel's ephemeral range (32768+ on Linux): the ranks' clients retry connect() until rank 0
    listens, and a port inside that range can be handed to one of them as its own source port in the meantime."""
    import random
    for _ in range(200):
        port = random.randint(20000, 29999)
        s = socket.socket()
        try:
            s.bind(("127.0.0.1", port))
        except OSError:
            continue
        finally:
            s.close()
        return port
    raise RuntimeError("no free port in 20000-29999")


def pair_similarity(g):
    """What the retrieval network would give: higher for pairs that share more scene (tests/test_scheduler.py)."""
    b = g["batch"]
    off = b["offsets"].astype(np.int64)
    n = np.diff(off)
    csum = np.concatenate([[0], np.cumsum(b["inlier"], dtype=np.int64)])
    ratio = (csum[off[1:]] - csum[off[:-1]]) / np.maximum(n, 1)   # == b["inlier"][a:z].mean() per pair
    i, j = g["pairs"][:, 0].astype(np.int64), g["pairs"][:, 1].astype(np.int64)
    return np.round(0.3 + 0.6 * ratio + 0.05 * ((i * 7 + j) % 3), 3)


def write_scene(path, g, wave, sim_kind, sim=None):
    """u32 V, P, wave, simKind | [V x V f64 similarity if simKind == 1] | per pair: u32 src, dst, n; f64 thr, similarity;
    n x 4 f64 rows (cv::Mat N x 4 CV_64F, the reference's correspondence matrix)."""
    b, V = g["batch"], len(g["R_gt"])
    sim = pair_similarity(g) if sim is None else np.asarray(sim, np.float64)   # (sim: a caller's own per-pair values)
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


def write_scene_bulk(path, g, wave, sim_kind=2):
    """The same scene in the BULK layout (simKind | 0x100) for graphs of 10^5 pairs: u32 V, P, wave, simKind | [dense
    similarity if simKind == 1] | u32 src[P], dst[P], n[P]; f64 thr[P], similarity[P] | every row as four f32 (x1 y1 x2 y2),
    pair after pair.  The values are f32-representable (the generators round to f32), so the driver's widening to the
    reference's CV_64F rows is exact; the file is half the size and is written by five array dumps instead of a Python loop."""
    b, V = g["batch"], len(g["R_gt"])
    sim = pair_similarity(g)
    P = len(g["pairs"])
    with open(path, "wb") as f:
        f.write(struct.pack("<IIII", V, P, wave, sim_kind | 0x100))
        if sim_kind == 1:
            dense = np.zeros((V, V))
            dense[g["pairs"][:, 0], g["pairs"][:, 1]] = sim
            dense[g["pairs"][:, 1], g["pairs"][:, 0]] = sim
            f.write(dense.astype("<f8").tobytes())
        g["pairs"][:, 0].astype("<u4").tofile(f)
        g["pairs"][:, 1].astype("<u4").tofile(f)
        np.diff(b["offsets"].astype(np.int64)).astype("<u4").tofile(f)
        np.full(P, 7.5e-4, "<f8").tofile(f)
        sim.astype("<f8").tofile(f)
        step = 1 << 22
        for a in range(0, len(b["x1"]), step):
            np.stack([b[k][a:a + step] for k in ("x1", "y1", "x2", "y2")], 1).astype("<f4").tofile(f)


def run_ranks(cmd, world, timeout=1500, extra_env=None):
    """One child process per rank (torch.distributed.run-style environment); returns their stdouts."""
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(extra_env or {})
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
    if os.environ.get("PGI_SHOW_STDERR"):  # diagnosis: the ranks' stderr (PGI_HOST_TIMING / PGI_ROTAVG_TIMING lines)
        import sys
        for _, _, e in outs:
            sys.stderr.write(e)
    return [o for _, o, _ in outs]


def rotations_of(blob, V):
    """the V global rotations every result file ends with"""
    return np.frombuffer(blob[-V * 72:], "<f8").reshape(V, 3, 3)


def align_error_deg(R, Rgt):
    """Per-view angular error after the gauge alignment R_k ~ Rgt_k G (G from view 0)."""
    G = Rgt[0].T @ R[0]
    d = np.einsum("kij,jl,kml->kim", Rgt, G, R)
    c = (np.trace(d, axis1=1, axis2=2) - 1) / 2
    return np.degrees(np.arccos(np.clip(c, -1, 1)))


def read_shard(blob, P):
    """mode "shard": u64 {pairs, graph edges, rotavg iterations, edges used} | P edge records | rotations"""
    hdr = struct.unpack_from("<4Q", blob, 0)
    return hdr, np.frombuffer(blob, EDGE_DTYPE, P, 32)


WAVES_KEYS = ("pairs_processed", "edges_added", "paths_searched", "paths_found", "touched_nodes", "poses_from_guess", "hypotheses",
              "waves", "graph_edges", "rotavg_iterations", "rotavg_edges_used", "stat_touched_nodes", "quirk_only_guesses")


def read_waves(blob):
    """modes "waves" / "waves_guided": u64 statistics | per graph edge {u32 src, dst; f64 score; R[9]; t[3]} | rotations"""
    st = struct.unpack_from("<%dQ" % len(WAVES_KEYS), blob, 0)
    n = st[8]
    rec = np.dtype([("src", "<u4"), ("dst", "<u4"), ("score", "<f8"), ("R", "<f8", 9), ("t", "<f8", 3)])
    return dict(zip(WAVES_KEYS, st)), np.frombuffer(blob, rec, n, 8 * len(WAVES_KEYS))


def _repetitions(stdout):
    """[(graph seconds, rotation-averaging seconds)] per repetition of the driver (PGI_DRIVER_REPS), and the index of the one
    reported: the median of the warm repetitions (all but the first, which allocates) by total time"""
    import re
    reps = []
    for ln in stdout.splitlines():
        if "seconds:" in ln:
            vals = re.findall(r"(?<![\w.])\d+\.\d+(?![\w.])", ln.split("seconds:")[1])
            reps.append((float(vals[0]), float(vals[1])))
    warm = list(range(1, len(reps))) if len(reps) > 1 else [0]
    k = sorted(warm, key=lambda i: reps[i][0] + reps[i][1])[(len(warm) - 1) // 2]
    return reps, k


def seconds_of(stdout):
    """the driver's own wall clock: (graph seconds, rotation-averaging seconds) of the median warm repetition"""
    reps, k = _repetitions(stdout)
    return reps[k]


def all_seconds_of(stdout):
    """graph + rotation-averaging seconds of every repetition, in order"""
    return [round(a + b, 4) for a, b in _repetitions(stdout)[0]]


def stages_of(stdout):
    """the driver's `stages:` line of the repetition seconds_of reports: {RunningStatistics time key: seconds}"""
    import re
    lines = [ln for ln in stdout.splitlines() if ln.startswith("stages:")]
    if not lines:
        return {}
    reps, k = _repetitions(stdout)
    line = lines[k] if len(lines) == len(reps) else lines[-1]
    return {k2.strip(): float(v) for k2, v in re.findall(r"([^=;]+)=([0-9.]+);", line[len("stages:"):])}


def all_stages_of(stdout):
    """the `stages:` line of EVERY repetition, in order: [{RunningStatistics time key: seconds}] (a slow
    repetition must show which stage carried the excess, not only the reported one)"""
    import re
    return [{k2.strip(): round(float(v), 4) for k2, v in re.findall(r"([^=;]+)=([0-9.]+);", ln[len("stages:"):])}
            for ln in stdout.splitlines() if ln.startswith("stages:")]


# ---- feature-level scenes (tests/cpp/test_pipeline.cpp: PoseGraphBuilder::processFeatures) -------------------------------------
PIPELINE_EXE = os.path.join(PKG, "test_pipeline")
PIPELINE_KEYS = ("pairs_processed", "edges_added", "paths_searched", "paths_found", "touched_nodes", "poses_from_guess", "hypotheses",
                 "waves", "graph_edges", "matching_runs", "quick_matching_runs", "guided_matching_runs", "guided_matches_added",
                 "track_number", "too_few_matches", "quirk_only_guesses")


def write_feature_scene(path, views, cam, sim, pairs, wave):
    """u32 V, P, wave | V x V f64 similarity | per view: u32 K; f64 f, w, h; K x 2 f32 keypoints; K x 128 f32 descriptors |
    per pair: u32 src, dst; f64 similarity"""
    with open(path, "wb") as f:
        f.write(struct.pack("<III", len(views), len(pairs), wave))
        f.write(np.ascontiguousarray(sim, "<f8").tobytes())
        for v in views:
            f.write(struct.pack("<Iddd", len(v["xy"]), *cam))
            f.write(np.ascontiguousarray(v["xy"], "<f4").tobytes())
            f.write(np.ascontiguousarray(v["desc"], "<f4").tobytes())
        for i, j, s in pairs:
            f.write(struct.pack("<IId", int(i), int(j), float(s)))


def parse_pipeline(buf, n_modes):
    """per mode: 16 u64 statistics (PIPELINE_KEYS) | per graph edge {u32 src, dst; f64 score; R[9]; t[3]}"""
    pos, res = 0, []
    for _ in range(n_modes):
        st = struct.unpack_from("<16Q", buf, pos)
        pos += 128
        edges = {}
        for _e in range(st[8]):
            s, d, sc = struct.unpack_from("<IId", buf, pos)
            R = np.frombuffer(buf, "<f8", 9, pos + 16).reshape(3, 3)
            t = np.frombuffer(buf, "<f8", 3, pos + 88)
            pos += 112
            edges[(s, d)] = (sc, R, t)
        res.append((st, edges))
    assert pos == len(buf)
    return res


def pipeline_timings(stdout):
    """the driver's per-mode lines -> {mode: dict(seconds=..., stages={...}, all_seconds=[...], repetition=k)}.  With several
    repetitions of a mode the one reported is the MEDIAN of the warm ones (all but the first, which allocates): a single
    repetition now and then takes 0.3 s longer in the upload on a shared box, whichever position it has."""
    import re
    reps = {}
    cur = None
    for line in stdout.splitlines():
        m = re.match(r"mode (\d+): (\d+) pairs -> (\d+) edges in ([0-9.]+) s", line)
        if m:
            cur = int(m.group(1))
            reps.setdefault(cur, []).append({"seconds": float(m.group(4)), "pairs": int(m.group(2)), "edges": int(m.group(3)), "stages": {}})
        elif cur is not None and "seconds:" in line:
            for name, val in re.findall(r"([A-Za-z*+ ]+?) ([0-9.]+)(?:,|$)", line.split("seconds:")[1]):
                reps[cur][-1]["stages"][name.strip()] = float(val)
    out = {}
    for mode, rs in reps.items():
        warm = list(range(1, len(rs))) if len(rs) > 1 else [0]
        k = sorted(warm, key=lambda i: rs[i]["seconds"])[(len(warm) - 1) // 2]
        out[mode] = dict(rs[k], all_seconds=[r["seconds"] for r in rs], all_stages=[r["stages"] for r in rs], repetition=k)
    return out


def graph_mode_metrics(g, blob, mode, sec_graph, sec_avg, stages):
    """One driver run (tests/cpp/test_distributed.cpp) of a scene graph -> the figures bench.py and the scripts report:
    seconds, stage clocks, edges, per-edge AUC@5 over the real pairs, global rotation error after gauge alignment."""
    V, P = len(g["R_gt"]), len(g["pairs"])
    gerr = align_error_deg(rotations_of(blob, V), g["R_gt"])
    m = {"seconds": round(sec_graph + sec_avg, 4), "seconds_graph": round(sec_graph, 4),
         "seconds_rotation_averaging": round(sec_avg, 4) if mode != "shard" else round(stages.get("[Rotation averaging]", 0.0), 4),
         "pairs_per_s": round(P / max(sec_graph + (sec_avg if mode != "shard" else 0.0), 1e-9), 1),
         "stages_s": {k: round(v, 4) for k, v in stages.items()},
         "global_rot_err_deg_mean": round(float(gerr.mean()), 4), "global_rot_err_deg_median": round(float(np.median(gerr)), 4)}

    def angles(Ra, Rb):
        c = (np.einsum("eij,eij->e", Ra, Rb) - 1.0) / 2.0
        return np.degrees(np.arccos(np.clip(c, -1.0, 1.0)))
    real = int((~g["wrong"]).sum())
    if mode == "shard":
        hdr, ed = read_shard(blob, P)
        ok = ed["status"] == 1
        eerr = np.full(P, np.inf)
        eerr[ok] = angles(ed["R"][ok].reshape(-1, 3, 3), g["batch"]["R"][ok])
        m.update(edges=int(hdr[1]), rotavg_iterations=int(hdr[2]), edge_rot_err_auc_at_5deg=round(S.auc_at(eerr[~g["wrong"]], 5.0), 4))
    else:
        stt, ged = read_waves(blob)
        key = g["pairs"][:, 0].astype(np.int64) * V + g["pairs"][:, 1]
        order = np.argsort(key)
        e_of = order[np.searchsorted(key[order], ged["src"].astype(np.int64) * V + ged["dst"])]
        eerr = angles(ged["R"].reshape(-1, 3, 3), g["batch"]["R"][e_of])
        m.update(edges=int(stt["graph_edges"]), waves=int(stt["waves"]), paths_searched=int(stt["paths_searched"]),
                 touched_nodes=int(stt["touched_nodes"]), poses_from_guess=int(stt["poses_from_guess"]),
                 quirk_only_guesses=int(stt["quirk_only_guesses"]), hypotheses=int(stt["hypotheses"]),
                 rotavg_iterations=int(stt["rotavg_iterations"]), edges_off_by_more_than_5deg=int((eerr > 5.0).sum()),
                 edge_rot_err_auc_at_5deg=round(float(np.sum(5.0 - eerr[eerr < 5.0]) / (5.0 * real)), 4))
    return m
