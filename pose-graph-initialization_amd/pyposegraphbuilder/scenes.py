"""Scene-graph surrogates of BASELINE configs 3/4/5 driven through the C++ host layer (tests/cpp/test_distributed.cpp):
the scene file format, the one-process-per-rank launcher and the readers of the driver's result files.  Shared by
tests/test_distributed_gpu.py, bench.py (graph-level extras) and scripts/config45_bench.py."""
import os
import socket
import struct
import subprocess

import numpy as np

from . import synthetic as S
from ._lib import EDGE_DTYPE

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(PKG, "test_distributed")

SCENES = {
    # name: (V, k, make_scene_graph overrides, wave size)
    "v340": (340, 12, dict(median_corr=500, max_corr=3000), 512),                                   # Madrid-Metropolis-sized
    "v5000": (5000, 4, dict(median_corr=100, min_corr=60, max_corr=400, ring=3), 4096),  # Trafalgar-sized; ring edges keep it connected
}


def make_scene(name, seed=11):
    V, k, kw, wave = SCENES[name]
    return S.make_scene_graph(V, k=k, seed=seed, outlier_pair_frac=0.03, **kw), wave


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


def seconds_of(stdout):
    """the driver's own wall clock: (graph seconds, rotation-averaging seconds)"""
    tail = stdout.strip().split("seconds:")[-1]  # the last repetition (PGI_DRIVER_REPS) is the warm one
    vals = [float(tok) for tok in tail.replace(",", " ").split() if tok.replace(".", "", 1).isdigit()]
    return vals[0], vals[1]


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
    """the driver's per-mode lines -> {mode: dict(seconds=..., stages={...})} (the last repetition of a mode wins)"""
    import re
    out = {}
    cur = None
    for line in stdout.splitlines():
        m = re.match(r"mode (\d+): (\d+) pairs -> (\d+) edges in ([0-9.]+) s", line)
        if m:
            cur = int(m.group(1))
            out[cur] = {"seconds": float(m.group(4)), "pairs": int(m.group(2)), "edges": int(m.group(3)), "stages": {}}
        elif cur is not None and "seconds:" in line:
            for name, val in re.findall(r"([A-Za-z*+ ]+?) ([0-9.]+)(?:,|$)", line.split("seconds:")[1]):
                out[cur]["stages"][name.strip()] = float(val)
    return out
