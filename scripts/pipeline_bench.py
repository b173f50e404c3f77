"""Features -> pose graph end to end (PoseGraphBuilder::processFeatures through the C++ driver), three configurations
(+ the host tracklet store as a fourth).  Usage: pipeline_bench.py [V [wave]]  -- V views of ~8000 keypoints, all pairs;
pipeline_bench.py config3 [wave]  -- BASELINE config 3's surrogate at full size: 340 views x ~8000 keypoints, the 20 next
views of every view as candidates (6590 pairs).  Prints the driver's own timing lines (setup/upload included)."""
import os, struct, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-graph-initialization_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pyposegraphbuilder import synthetic as S
import scene_drivers as SC
full = len(sys.argv) > 1 and sys.argv[1] == "config3"
wave = int(sys.argv[2]) if len(sys.argv) > 2 else (512 if full else 128)
if full:
    views, poses, cam, sim, pairs = S.make_feature_scene(340, 8000, band=20)
    V = len(views)
else:
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    rng = np.random.default_rng(7)
    views, poses, cam = S.make_feature_views(rng, n_views=V, n_points=7000, n_clutter=2750, desc_noise=0.012)
    sim = np.zeros((V, V)); pairs = []
    for i in range(V):
        for j in range(i + 1, V):
            sim[i, j] = sim[j, i] = round(0.3 + 0.6 * rng.random(), 3)
            pairs.append((i, j, sim[i, j]))
with tempfile.TemporaryDirectory() as d:
    fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
    SC.write_feature_scene(fin, views, cam, sim, pairs, wave)
    print("views %d, keypoints/view ~%d, pairs %d, wave %d" % (V, np.mean([len(v["xy"]) for v in views]), len(pairs), wave))
    r = subprocess.run([os.path.join(ROOT, "pose-graph-initialization_amd", "test_pipeline"), fin, fout], capture_output=True, text=True)
    print(r.stdout, r.stderr[-6000:])
    if os.environ.get("PGI_PIPELINE_STDERR"):  # the driver's full stderr (e.g. with PGI_TRACKLETS_TIMING=1)
        open(os.environ["PGI_PIPELINE_STDERR"], "w").write(r.stderr)
    # the tracklet store in HBM (mode 2) and the host store (mode 3) must give the same graph, counter for counter
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_feature_pipeline import parse
    res = parse(open(fout, "rb").read())
    (st2, e2), (st3, e3) = res[2], res[3]
    same = st2 == st3 and e2.keys() == e3.keys() and all(
        e2[k][0] == e3[k][0] and np.array_equal(e2[k][1], e3[k][1]) and np.array_equal(e2[k][2], e3[k][2]) for k in e2)
    print("mode 2 == mode 3 (statistics, %d edges, scores, rotations, translations): %s; %d tracks" % (len(e2), same, st2[13]))
