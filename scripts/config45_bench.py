#!/usr/bin/env python3
"""BASELINE configs 4 and 5 end to end on ONE GPU (world 1), through the C++ host layer (tests/cpp/test_distributed.cpp):
  config45_bench.py [scene] [modes] [world]      scene: v5000 (default; SURVEY 8d's density: ~106 000 pairs, ~77 M rows),
                                                 v340, v5000_ring, v340_thin (tests/scene_drivers.py SCENES)
  shard         estimate all pairs + gather + L1/IRLS rotation averaging            (config 4)
  waves         A*-scheduled waves, reference-style pose guesses                    (config 5, reference guesses)
  waves_guided  the same with rotation-guided re-estimation                         (config 5 as BASELINE names it)
Prints the driver's own wall-clock and stage lines (scene generation and file I/O excluded); PGI_HOST_TIMING=1 adds the
phases of every estimatePoses call on stderr."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-graph-initialization_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from pyposegraphbuilder import synthetic as S
import scene_drivers as SC
name = sys.argv[1] if len(sys.argv) > 1 else "v5000"
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["shard", "waves", "waves_guided"]
world = int(sys.argv[3]) if len(sys.argv) > 3 else 1
t0 = time.time()
g, wave = SC.make_scene(name)
V, P = len(g["R_gt"]), len(g["pairs"])
print("scene %s: views %d, candidate pairs %d, rows %d, wave %d (generated in %.1f s)" % (
    name, V, P, int(g["batch"]["offsets"][-1]), wave, time.time() - t0), flush=True)
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "scene.bin")
    t0 = time.time()
    (SC.write_scene_bulk if SC.SCENES[name][4] else SC.write_scene)(path, g, wave, sim_kind=2)
    print("scene file %.2f GB written in %.1f s" % (os.path.getsize(path) / 1e9, time.time() - t0), flush=True)
    lut = {(int(i), int(j)): e for e, (i, j) in enumerate(g["pairs"])}
    for mode in modes:
        t0 = time.time()
        outs = SC.run_ranks([SC.EXE, path, os.path.join(d, mode), mode], world, extra_env={"PGI_DRIVER_REPS": os.environ.get("PGI_DRIVER_REPS", "2")})
        print(outs[0].strip(), "\n  (process wall %.1f s)" % (time.time() - t0), flush=True)
        blob = open(os.path.join(d, mode) + ".0", "rb").read()
        gerr = SC.align_error_deg(SC.rotations_of(blob, V), g["R_gt"])
        if mode == "shard":
            hdr, ed = SC.read_shard(blob, P)
            ok = ed["status"] == 1
            e_all = np.full(P, np.inf)
            e_all[ok] = [S.rot_err_deg(ed["R"][e].reshape(3, 3), g["batch"]["R"][e]) for e in np.nonzero(ok)[0]]
            print("  edges %d, AUC@5 %.4f, global rotation error mean %.4f median %.4f deg" % (
                hdr[1], S.auc_at(e_all[~g["wrong"]], 5.0), gerr.mean(), np.median(gerr)))
        else:
            st, ged = SC.read_waves(blob)
            eerr = np.array([S.rot_err_deg(r["R"].reshape(3, 3), g["batch"]["R"][lut[(int(r["src"]), int(r["dst"]))]]) for r in ged])
            print("  %s" % {k: int(v) for k, v in st.items()})
            print("  AUC@5 %.4f, edges off by > 5 deg %d, global rotation error mean %.4f median %.4f deg" % (
                float(np.sum(5.0 - eerr[eerr < 5.0]) / (5.0 * int((~g["wrong"]).sum()))), int((eerr > 5).sum()), gerr.mean(), np.median(gerr)))
