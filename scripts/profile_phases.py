#!/usr/bin/env python3
"""Per-phase cycle accounting of estimate_pose_kernel (instrumented libpgi_prof.so, s_memtime).
Reports wave-ticks per phase summed over all wavefronts, as a share of the total."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-graph-initialization_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pyposegraphbuilder import _lib as L
L.LIB_PATH = os.path.join(ROOT, "pose-graph-initialization_amd", "libpgi_prof.so")
import torch
from pyposegraphbuilder import Engine, synthetic as S

NAMES = {0: "stage rows -> LDS", 1: "sample + gather", 2: "nullspace 5x9 + MGS", 3: "quads + cubic rows",
         4: "Gauss-Jordan 10x20", 5: "B(z), det poly", 6: "bracket grid", 7: "Newton refine",
         8: "E from root + orientation", 9: "enqueue", 10: "score models", 11: "LO entry", 12: "LO normal matrix",
         13: "LO jacobi 9x9", 14: "LO quads+rows", 15: "LO Gauss-Jordan", 16: "LO B/det", 17: "LO brackets",
         18: "LO refine", 19: "LO E from root", 20: "LO wait for wave 0", 21: "LO score+combine",
         22: "round barrier wait", 23: "merge/termination", 24: "epilogue (mask+decompose)"}
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = sys.argv[2] if len(sys.argv) > 2 else "2000"   # a row count, or a scene name (v5000: the dense scene's own rows, first P pairs;
fb = int(sys.argv[3]) if len(sys.argv) > 3 else 0   #  "v5000:1344" keeps only pairs of at most 1344 rows -- one size class)
eng = Engine(fixed_budget=fb)
if N[0] == "v":
    import scene_drivers as SC
    name, _, cap = N.partition(":")
    g, _w = SC.make_scene(name)
    bb = g["batch"]
    off = np.asarray(bb["offsets"], np.int64)
    n_all = np.diff(off)
    keep = np.nonzero(n_all <= int(cap))[0][:P] if cap else np.arange(min(P, len(n_all)))
    P = len(keep)
    idx = np.concatenate([np.arange(off[k], off[k + 1]) for k in keep])
    b = {k: bb[k][idx] for k in ("x1", "y1", "x2", "y2")}
    b["offsets"] = np.concatenate([[0], np.cumsum(n_all[keep])]).astype(np.uint64)
    print("rows of scene %s: %d pairs, median %d rows" % (N, P, np.median(n_all[keep])))
else:
    N = int(N)
    b = S.make_batch(np.arange(P), N)
db = eng.upload(b["x1"], b["y1"], b["x2"], b["y2"], b["offsets"], 7.5e-4, seed=1)
buf = torch.zeros(32 + 12, dtype=torch.int64, device=eng.device)
eng.estimate_pose_batch(db); torch.cuda.synchronize()
eng._lib.pgi_internal_set_profile_buffer.argtypes = [C.c_void_p, C.c_void_p]
eng._lib.pgi_internal_set_profile_buffer(eng._ctx, C.c_void_p(buf.data_ptr()))
a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); e, m = eng.estimate_pose_batch(db); z.record(); torch.cuda.synchronize()
raw = buf.cpu().numpy()
acc = raw[:32].astype(np.float64)
tot = acc.sum()
got = eng.edges_to_numpy(e)
print("P=%d N=%s budget=%d  kernel %.3f ms  hyps=%.1f lo=%.2f  (s_memtime ticks, summed over all wavefronts; PGI_K1_NW=%s)" % (
    P, N, fb, a.elapsed_time(z), got["iters"].mean(), got["lo_runs"].mean(), os.environ.get("PGI_K1_NW", "auto")))
for i in range(25):
    print("%2d %-28s %14.0f  %6.2f%%  %10.0f ticks/pair" % (i, NAMES.get(i, ""), acc[i], 100 * acc[i] / tot, acc[i] / P))
# slot occupancy per rows variant (one launch each): the workgroups' residence on the 100 MHz wall clock.  A persistent workgroup
# leaves when its class's list is dry, so (1 - residence / span) is the share of slot time the launch's wind-down leaves empty.
occ = raw[32:].view(np.uint64).reshape(3, 4)
used = [i for i in range(3) if occ[i, 3]]
if not used:
    print("(no occupancy counters: libpgi_prof.so predates round 6 -- rebuild it: make -C pose-graph-initialization_amd libpgi_prof.so)")
    sys.exit(0)
t_first = min(int(~occ[i, 2] & 0xFFFFFFFFFFFFFFFF) for i in used)
for i in used:
    start, end = int(~occ[i, 2] & 0xFFFFFFFFFFFFFFFF), int(occ[i, 1])
    span = (end - start) / 1e5
    print("rows variant %d: %6d workgroups, first start %8.3f ms, last exit %8.3f ms (span %7.3f ms), mean residence %7.3f ms = %.1f %% of the span" % (
        i, occ[i, 3], (start - t_first) / 1e5, (end - t_first) / 1e5, span, occ[i, 0] / occ[i, 3] / 1e5,
        100.0 * occ[i, 0] / occ[i, 3] / 1e5 / max(span, 1e-9)))
