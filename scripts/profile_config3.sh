# config 3 from features at full size under rocprofv3 (kernel trace + stats): which kernels make up the 0.18 s?
# Usage (GPU box): bash scripts/profile_config3.sh [modes, default 2] [tag, default r05]
M=${1:-2}
T=${2:-r05}
set -x
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
PGI_DRIVER_REPS=2 rocprofv3 --kernel-trace --stats -d gpurun_out/${T}_config3_trace -o $T -- pose-graph-initialization_amd/test_pipeline /tmp/config3_features.bin /tmp/config3_features.out $M > gpurun_out/${T}_config3_trace.log 2>&1
python3 scripts/rocpd_summary.py $(find gpurun_out/${T}_config3_trace -name "*.db" | head -1) > gpurun_out/${T}_config3_trace_summary.txt 2>&1
head -40 gpurun_out/${T}_config3_trace_summary.txt | cut -c1-150; cat gpurun_out/${T}_config3_trace.log | tail -5
rm -rf gpurun_out/${T}_config3_trace
rm -f /tmp/config3_features.bin /tmp/config3_features.out
