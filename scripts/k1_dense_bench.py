#!/usr/bin/env python3
"""K1 alone on the rows of the dense V = 5000 scene (BASELINE configs 4/5 at SURVEY 8d's density: ~106 000 pairs, ~77 M rows,
median 588 rows per pair), resident in HBM, ONE estimate_pose_batch call per repetition -- the floor of config 4's
estimation stage.  Several builds of libpgi (paths relative to the package) are timed in one process, interleaved, and
their result bytes compared -- or, "nw=1" / "nw=2,overlap=0" / "nw=0" ..., the default build with these launcher settings
(nw: wavefronts per pair, 0 = the launcher's own rule; minwgs / hybrid / overlap / persistent: PGI_LDS_MIN_WGS, PGI_HYBRID_ROWS,
PGI_CLASS_OVERLAP, PGI_K1_PERSISTENT -- all read when the context is made):
    k1_dense_bench.py [libpgi.so nw=1 nw=2,overlap=0 nw=4 nw=0 ...]
Environment: K1D_SCENE (v5000), K1D_ROUNDS (5), K1D_PARAMS ("round_size=16,lo_iters=2": pgi_params overrides for every build),
K1D_MAXPAIRS (all): keep only the first so-many pairs (profiling runs)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-graph-initialization_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pyposegraphbuilder import _lib as L, synthetic as S
import scene_drivers as SC
from pyposegraphbuilder.engine import Engine

libs = sys.argv[1:] or ["libpgi.so"]
SPEC_ENV = {"nw": "PGI_K1_NW", "minwgs": "PGI_LDS_MIN_WGS", "hybrid": "PGI_HYBRID_ROWS", "overlap": "PGI_CLASS_OVERLAP", "persistent": "PGI_K1_PERSISTENT"}
name = os.environ.get("K1D_SCENE", "v5000")
t0 = time.time()
g, _ = SC.make_scene(name)
b = g["batch"]
off = np.asarray(b["offsets"], np.int64)
maxp = int(os.environ.get("K1D_MAXPAIRS", "0"))
if maxp and maxp < len(off) - 1:
    off = off[:maxp + 1]
    for k in ("x1", "y1", "x2", "y2"):
        b[k] = b[k][:off[-1]]
n = np.diff(off)
P = len(n)
print("scene %s: %d pairs, %d rows (median %d, mean %.0f, max %d), generated in %.1f s" % (
    name, P, off[-1], np.median(n), n.mean(), n.max(), time.time() - t0), flush=True)
for cap in (640, 1088, 1344, 1920, 2176, 3904):
    print("  pairs with <= %4d rows: %6d (%.1f %%), their rows %.1f %%" % (cap, (n <= cap).sum(), 100.0 * (n <= cap).mean(),
                                                                          100.0 * n[n <= cap].sum() / n.sum()))
params = {}
for kv in filter(None, os.environ.get("K1D_PARAMS", "").split(",")):
    k, v = kv.split("=")
    params[k] = float(v) if "." in v else int(v)
engs = []
for path in libs:
    L._lib = None
    for v in SPEC_ENV.values():
        os.environ.pop(v, None)
    lib_name, _, spec = path.rpartition(":") if ":" in path else (("", "", path) if "=" in path else (path, "", ""))
    for kv in filter(None, spec.split(",")):  # "nw=1,overlap=0" (or "libpgi_x.so:nw=1"): launcher settings, read when the context is made
        k, v = kv.split("=")
        os.environ[SPEC_ENV[k]] = v
    L.LIB_PATH = os.path.join(ROOT, "pose-graph-initialization_amd", lib_name or "libpgi.so")
    e = Engine()
    for v in SPEC_ENV.values():
        os.environ.pop(v, None)
    if params:
        e.set_params(**params)
    engs.append(e)
dbs = [e.upload(b["x1"], b["y1"], b["x2"], b["y2"], off.astype(np.uint64), 7.5e-4, seed=0x5EED) for e in engs]
res = {i: [] for i in range(len(engs))}
outs, stats = {}, {}
for rnd in range(int(os.environ.get("K1D_ROUNDS", "5")) + 1):
    for i, e in enumerate(engs):
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ed, m = e.estimate_pose_batch(dbs[i]); z.record(); torch.cuda.synchronize()
        if rnd:
            res[i].append(a.elapsed_time(z))
        else:
            got = e.edges_to_numpy(ed)
            outs[i] = ed.cpu().numpy().tobytes() + m.cpu().numpy().tobytes()
            ok = got["status"] == 1
            err = np.full(P, np.inf)
            err[ok] = [S.rot_err_deg(got["R"][k].reshape(3, 3), b["R"][k]) for k in np.nonzero(ok)[0]]
            good = ~g["wrong"][:P]
            stats[i] = (got["iters"].mean(), got["lo_runs"].mean(), int(ok.sum()), S.auc_at(err[good], 5.0))
for i in res:
    ms = np.array(res[i])
    print("%-18s %s  median %.3f ms = %.3f M pairs/s, %.2f G rows/s | hyps %.1f, refits %.2f, edges %d, AUC@5 %.4f | identical to first: %s" % (
        libs[i], " ".join("%.2f" % v for v in ms), np.median(ms), P / np.median(ms) / 1e3, off[-1] / np.median(ms) / 1e6,
        stats[i][0], stats[i][1], stats[i][2], stats[i][3], outs[i] == outs[0]))
