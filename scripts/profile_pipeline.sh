cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 - <<'PY'
import os, struct, sys, numpy as np
sys.path.insert(0, "pose-graph-initialization_amd")
from pyposegraphbuilder import synthetic as S
V, wave = 40, 64
rng = np.random.default_rng(7)
views, poses, cam = S.make_feature_views(rng, n_views=V, n_points=7000, n_clutter=2750, desc_noise=0.012)
sim = np.zeros((V, V)); pairs = []
for i in range(V):
    for j in range(i + 1, V):
        sim[i, j] = sim[j, i] = round(0.3 + 0.6 * rng.random(), 3)
        pairs.append((i, j, sim[i, j]))
with open("/tmp/pipe_in.bin", "wb") as f:
    f.write(struct.pack("<III", V, len(pairs), wave))
    f.write(sim.astype("<f8").tobytes())
    for v in views:
        f.write(struct.pack("<Iddd", len(v["xy"]), *cam))
        f.write(v["xy"].astype("<f4").tobytes()); f.write(v["desc"].astype("<f4").tobytes())
    for i, j, s in pairs:
        f.write(struct.pack("<IId", i, j, s))
PY
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/pipe_tr -o pipe -- pose-graph-initialization_amd/test_pipeline /tmp/pipe_in.bin /tmp/pipe_out.bin > /dev/null 2>&1
python3 scripts/rocpd_summary.py $(find gpurun_out/pipe_tr -name "*.db" | head -1) 2>&1 | cut -c1-120 | head -22
