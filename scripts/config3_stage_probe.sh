# config 3 from features at full size, stage split of the C++ driver (two repetitions inside the process; the second is warm).
# Usage (GPU box): bash scripts/config3_stage_probe.sh [modes, default 4]
M=${1:-4}
python3 - <<'PY'
import sys
sys.path.insert(0, "pose-graph-initialization_amd")
sys.path.insert(0, "tests")
from pyposegraphbuilder import synthetic as S
import scene_drivers as SC
views, poses, cam, sim, pairs = S.make_feature_scene(340, 8000, band=20)
SC.write_feature_scene("/tmp/config3_features.bin", views, cam, sim, pairs, 512)
PY
PGI_DRIVER_REPS=${REPS:-2} timeout 120 pose-graph-initialization_amd/test_pipeline /tmp/config3_features.bin /tmp/config3_features.out $M
rm -f /tmp/config3_features.bin /tmp/config3_features.out
