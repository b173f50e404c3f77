# how full the two-kernel scan's arena gets on config 3 from features (PGI_GUIDED_ARENA_REPORT=1)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python3 - <<'PY'
import sys
sys.path.insert(0, "pose-graph-initialization_amd"); sys.path.insert(0, "tests")
from pyposegraphbuilder import synthetic as S
import scene_drivers as SC
views, poses, cam, sim, pairs = S.make_feature_scene(340, 8000, band=20)
SC.write_feature_scene("/tmp/config3_features.bin", views, cam, sim, pairs, 512)
PY
PGI_GUIDED_ARENA_REPORT=1 pose-graph-initialization_amd/test_pipeline /tmp/config3_features.bin /tmp/config3_features.out 4 2>&1 | grep "arena" | tail -n 4
