cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out
cd $R
python3 - <<'PY'
import sys
sys.path.insert(0, "pose-graph-initialization_amd")
sys.path.insert(0, "tests")
from pyposegraphbuilder import synthetic as S
import scene_drivers as SC
views, poses, cam, sim, pairs = S.make_feature_scene(340, 8000, band=20)
SC.write_feature_scene("/tmp/config3_features.bin", views, cam, sim, pairs, 512)
PY
for v in ${*:-PGI_GUIDED_LANES=2 PGI_GUIDED_CAP=16 PGI_GUIDED_CAP=20}; do
  export $v
  rm -rf gpurun_out/v3
  rocprofv3 --kernel-trace -d gpurun_out/v3 -o v -- pose-graph-initialization_amd/test_pipeline /tmp/config3_features.bin /tmp/config3_features.out 4 > gpurun_out/v3.log 2>&1
  echo "== $v: $(grep -i "seconds\|run " gpurun_out/v3.log | tail -n 1 | cut -c1-150)"
  python3 scripts/rocpd_summary.py $(find gpurun_out/v3 -name "*.db" | head -1) 2>&1 | grep "guided_scan\|guided_angle\|guided_gate\|guided_sums" | cut -c1-130
  rm -rf gpurun_out/v3
  unset PGI_GUIDED_LANES PGI_GUIDED_CAP
done
