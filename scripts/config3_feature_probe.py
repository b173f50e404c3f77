import os, sys, subprocess, tempfile, numpy as np
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, os.path.join(ROOT, "pose-graph-initialization_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pyposegraphbuilder import synthetic as S
import scene_drivers as SC
views, poses, cam, sim, pairs = S.make_feature_scene(340, 8000, band=20)
with tempfile.TemporaryDirectory() as d:
    fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
    SC.write_feature_scene(fin, views, cam, sim, pairs, 512)
    r = subprocess.run([SC.PIPELINE_EXE, fin, fout, "0124"], capture_output=True, text=True)
    print(r.stdout, r.stderr[-2000:])
    res = SC.parse_pipeline(open(fout, "rb").read(), 4)
for name, (st, e) in zip(("plain", "A*", "A*+hashing", "A*+hashing guided"), res):
    k = dict(zip(SC.PIPELINE_KEYS, st))
    err = np.array([S.rot_err_deg(e[key][1], poses[key[1]][0] @ poses[key[0]][0].T) for key in e])
    gap = np.array([key[1] - key[0] for key in e])
    print("%-18s edges %d guesses %d quirk-only %d hyps %d | err<0.5 %.3f <1 %.3f <5 %.3f | bad(>5deg) by gap: near(<=7) %.3f mid %.3f far(>14) %.3f" % (
        name, len(e), k["poses_from_guess"], k["quirk_only_guesses"], k["hypotheses"], np.mean(err < .5), np.mean(err < 1), np.mean(err < 5),
        np.mean(err[gap <= 7] > 5), np.mean(err[(gap > 7) & (gap <= 14)] > 5), np.mean(err[gap > 14] > 5)))
