cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python3 - <<'PY'
import sys
sys.path.insert(0, "pose-graph-initialization_amd"); sys.path.insert(0, "tests")
from pyposegraphbuilder import synthetic as S
import scene_drivers as SC
views, poses, cam, sim, pairs = S.make_feature_scene(340, 8000, band=20)
SC.write_feature_scene("/tmp/config3_features.bin", views, cam, sim, pairs, 512)
PY
export PGI_GUIDED_SPLIT=1
for a in 0 1 2 4 6; do
  export PGI_ABL=$a
  rm -rf gpurun_out/v3
  rocprofv3 --kernel-trace -d gpurun_out/v3 -o v -- pose-graph-initialization_amd/test_pipeline /tmp/config3_features.bin /tmp/config3_features.out 4 > gpurun_out/v3.log 2>&1
  echo "== ABL $a: $(python3 scripts/rocpd_summary.py $(find gpurun_out/v3 -name "*.db" | head -1) 2>&1 | grep "guided_deal\|guided_sum" | cut -c28-100 | tr '\n' '|')"
  rm -rf gpurun_out/v3
done
