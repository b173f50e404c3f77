"""The scene-graph surrogates of BASELINE configs 3/4/5 at SURVEY 8d's density (pyposegraphbuilder/synthetic.py
make_scene_graph_dense, tests/scene_drivers.py): generator properties, the bulk scene file, the stage-line parsers.  CPU."""
import struct

import numpy as np

from pyposegraphbuilder import synthetic as S
import scene_drivers as SC


def small():
    return S.make_scene_graph_dense(120, k=12, seed=3, median_corr=200, min_corr=60, max_corr=900, block_pairs=97)


def test_dense_scene_is_deterministic_and_well_formed():
    g, h = small(), small()   # blocks run on a thread pool: the result must not depend on their schedule
    for key in ("x1", "y1", "x2", "y2", "inlier"):
        assert np.array_equal(g["batch"][key], h["batch"][key]), key
    assert np.array_equal(g["pairs"], h["pairs"]) and np.array_equal(g["sizes"], h["sizes"])
    P = len(g["pairs"])
    assert (g["pairs"][:, 0] < g["pairs"][:, 1]).all() and len({tuple(p) for p in g["pairs"]}) == P      # each pair once, src < dst
    deg = np.bincount(g["pairs"].ravel(), minlength=120)
    assert deg.min() >= 12                                       # every view has its k nearest views
    assert g["sizes"].min() >= 60 and g["sizes"].max() <= 900 and int(g["batch"]["offsets"][-1]) == int(g["sizes"].sum())
    b = g["batch"]
    assert np.isfinite(b["x1"]).all() and np.isfinite(b["y2"]).all()
    # wrongly retrieved pairs carry no inlier; the others an inlier share inside the requested band (Bernoulli per row)
    off = b["offsets"].astype(np.int64)
    ratio = np.array([b["inlier"][off[e]:off[e + 1]].mean() for e in range(P)])
    assert (ratio[g["wrong"]] == 0).all() and (ratio[~g["wrong"]] > 0.2).all() and (ratio[~g["wrong"]] < 0.95).all()
    # the inlier rows satisfy the pair's epipolar geometry to noise level: x2^T [t]x R x1 ~ 0
    e = int(np.nonzero(~g["wrong"])[0][0])
    a, z = off[e], off[e + 1]
    E = np.cross(np.eye(3), b["t"][e]) @ b["R"][e]
    x1 = np.stack([b["x1"][a:z], b["y1"][a:z], np.ones(z - a)], 1)
    x2 = np.stack([b["x2"][a:z], b["y2"][a:z], np.ones(z - a)], 1)
    res = np.abs(np.einsum("ni,ij,nj->n", x2, E, x1))
    assert np.median(res[b["inlier"][a:z]]) < 2e-3 and np.median(res[~b["inlier"][a:z]]) > 1e-2


def test_pair_similarity_equals_the_per_pair_loop():
    g = small()
    b = g["batch"]
    loop = np.array([round(0.3 + 0.6 * b["inlier"][int(b["offsets"][e]):int(b["offsets"][e + 1])].mean() + 0.05 * ((int(i) * 7 + int(j)) % 3), 3)
                     for e, (i, j) in enumerate(g["pairs"])])
    assert np.array_equal(SC.pair_similarity(g), loop)


def test_bulk_scene_file_layout(tmp_path):
    g = small()
    path = str(tmp_path / "scene.bin")
    SC.write_scene_bulk(path, g, 64, sim_kind=2)
    blob = open(path, "rb").read()
    V, P, wave, kind = struct.unpack_from("<IIII", blob, 0)
    assert (V, P, wave, kind) == (120, len(g["pairs"]), 64, 2 | 0x100)
    pos = 16
    src = np.frombuffer(blob, "<u4", P, pos); pos += 4 * P
    dst = np.frombuffer(blob, "<u4", P, pos); pos += 4 * P
    n = np.frombuffer(blob, "<u4", P, pos); pos += 4 * P
    thr = np.frombuffer(blob, "<f8", P, pos); pos += 8 * P
    sim = np.frombuffer(blob, "<f8", P, pos); pos += 8 * P
    rows = np.frombuffer(blob, "<f4", 4 * int(n.sum()), pos).reshape(-1, 4)
    assert pos + rows.nbytes == len(blob)
    assert np.array_equal(src, g["pairs"][:, 0]) and np.array_equal(dst, g["pairs"][:, 1]) and np.array_equal(n, g["sizes"])
    assert (thr == 7.5e-4).all() and np.array_equal(sim, SC.pair_similarity(g))
    b = g["batch"]
    assert np.array_equal(rows, np.stack([b["x1"], b["y1"], b["x2"], b["y2"]], 1))


def test_take_pairs_is_a_consistent_sample():
    g = small()
    idx = np.array([5, 0, 17])
    s = S.take_pairs(g, idx)
    b = g["batch"]
    for q, e in enumerate(idx):
        a, z = int(b["offsets"][e]), int(b["offsets"][e + 1])
        sa, sz = int(s["offsets"][q]), int(s["offsets"][q + 1])
        assert sz - sa == z - a and np.array_equal(s["x2"][sa:sz], b["x2"][a:z]) and np.array_equal(s["R"][q], b["R"][e])


def test_driver_output_parsers():
    out = ("rank 0/1 transport none mode shard edges 15011 rotavg iters 10 | seconds: estimate + gather + average 0.0600, rotation averaging 0.0000\n"
           "stages: [A*]=0.0123; [Pose estimation]=0.0450; [Pose estimation] convert + upload + launch (chunks)=0.0012;\n"
           "rank 0/1 transport none mode shard edges 15011 rotavg iters 10 | seconds: estimate + gather + average 0.0500, rotation averaging 0.0100\n"
           "stages: [A*]=0.0100; [Rotation averaging]=0.0200;\n")
    assert SC.seconds_of(out) == (0.05, 0.01)                          # two repetitions: the warm one
    assert SC.stages_of(out) == {"[A*]": 0.01, "[Rotation averaging]": 0.02}
    # four repetitions: the median of the three warm ones (an outlier, wherever it falls, is listed but not reported)
    more = out + ("rank 0/1 ... | seconds: estimate + gather + average 0.4000, rotation averaging 0.0100\nstages: [A*]=0.3000;\n"
                  "rank 0/1 ... | seconds: estimate + gather + average 0.0550, rotation averaging 0.0100\nstages: [A*]=0.0150;\n")
    assert SC.seconds_of(more) == (0.055, 0.01) and SC.stages_of(more) == {"[A*]": 0.015}
    assert SC.all_seconds_of(more) == [0.06, 0.06, 0.41, 0.065]
    pipe = ("mode 4: 10 pairs -> 9 edges in 0.300 s (x)\n        seconds: upload + prepare 0.100, A* 0.004\n"
            "mode 4: 10 pairs -> 9 edges in 0.171 s (x)\n        seconds: upload + prepare 0.027, A* 0.004\n"
            "mode 4: 10 pairs -> 9 edges in 0.491 s (x)\n        seconds: upload + prepare 0.343, A* 0.004\n"
            "mode 4: 10 pairs -> 9 edges in 0.175 s (x)\n        seconds: upload + prepare 0.029, A* 0.004\n")
    t = SC.pipeline_timings(pipe)[4]
    assert t["seconds"] == 0.175 and t["repetition"] == 3 and t["stages"]["upload + prepare"] == 0.029
    assert t["all_seconds"] == [0.3, 0.171, 0.491, 0.175]


def test_every_repetitions_stage_clocks_are_parsed():
    """bench.py puts the stage clocks of EVERY repetition into its line (round 5: a slow repetition must name its stage)."""
    out = ("rank 0/1 transport none mode shard edges 5 rotavg iters 8 | seconds: estimate + gather + average 0.2000, rotation averaging 0.0000\n"
           "stages: [Pose estimation]=0.1500; [Pose estimation] download=0.0010; [Rotation averaging]=0.0160;\n"
           "rank 0/1 transport none mode shard edges 5 rotavg iters 8 | seconds: estimate + gather + average 0.1000, rotation averaging 0.0000\n"
           "stages: [Pose estimation]=0.0800; [Pose estimation] download=0.0009; [Rotation averaging]=0.0150;\n"
           "rank 0/1 transport none mode shard edges 5 rotavg iters 8 | seconds: estimate + gather + average 0.1100, rotation averaging 0.0000\n"
           "stages: [Pose estimation]=0.0900; [Pose estimation] download=0.0011; [Rotation averaging]=0.0155;\n")
    assert SC.all_seconds_of(out) == [0.2, 0.1, 0.11]
    st = SC.all_stages_of(out)
    assert len(st) == 3 and st[1] == {"[Pose estimation]": 0.08, "[Pose estimation] download": 0.0009, "[Rotation averaging]": 0.015}
    assert SC.seconds_of(out) == (0.1, 0.0)                              # the median of the warm repetitions (here: the lower of two)
    assert SC.stages_of(out)["[Pose estimation]"] == 0.08                # ... and its own stage line
    pipe = ("mode 2: 10 pairs -> 9 edges in 0.300 s (33.3 pairs/s; 1 matched, 2 quick, 3 guided runs)\n"
            "        seconds: upload + prepare 0.200, quick matching 0.004, matching 0.030, correspondences 0.001, A* 0.004, pose estimation 0.017, guided 0.048, commit + tracklets 0.030\n"
            "mode 2: 10 pairs -> 9 edges in 0.170 s (58.8 pairs/s; 1 matched, 2 quick, 3 guided runs)\n"
            "        seconds: upload + prepare 0.027, quick matching 0.004, matching 0.030, correspondences 0.001, A* 0.004, pose estimation 0.017, guided 0.048, commit + tracklets 0.030\n")
    t = SC.pipeline_timings(pipe)[2]
    assert t["all_seconds"] == [0.3, 0.17] and t["seconds"] == 0.17
    assert [s["upload + prepare"] for s in t["all_stages"]] == [0.2, 0.027]


def test_bench_summary_and_amdahl_helpers():
    """bench.py's last key (`summary`, < 1 KB: what the driver's truncated record must still show) and the Amdahl ceilings of a
    leg whose replicated host seconds every rank repeats."""
    import importlib.util
    import json
    import os
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(SC.ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    a = bench.amdahl(0.12, 0.03)
    assert a == {"n2": round(0.12 / (0.03 + 0.045), 2), "n4": round(0.12 / (0.03 + 0.0225), 2), "n8": round(0.12 / (0.03 + 0.01125), 2)}
    assert bench.amdahl(0.1, 0.0)["n8"] == 8.0 and bench.amdahl(0.1, 0.2)["n8"] == 1.0
    line = json.load(open(os.path.join(SC.ROOT, "profiles", "r06_bench_full.json")))   # the full record of the round's final run
    s = bench.compact_summary({k: v for k, v in line.items() if k != "summary"})
    assert s == line["summary"] and len(json.dumps(s)) < 2048
    # ... and the line the driver parses is built from it (tests/test_bench_line.py holds the size and key checks)
    assert json.loads(bench.compact_line(line))["summary"] == bench._short(s)
    assert list(line)[-1] == "summary"                                   # the last key of the line
    for key in ("k1_edges_per_s", "k1_frac_hbm", "k2_frac_hbm", "match_exact_frac_mfma", "config3_s", "config4_s", "config4_cpu_parity",
                "config5_guided_s", "worst_over_median_repetition"):
        assert s[key] is not None, key
