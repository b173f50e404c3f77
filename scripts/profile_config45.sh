# configs 4 and 5 on the dense V = 5000 scene under rocprofv3 (kernel trace + stats): which kernels make up the 0.10 / 0.15 s?
# Usage (GPU box): bash scripts/profile_config45.sh
T=${1:-r05}
set -x
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out
cd $R
python3 - <<'PY'
import sys
sys.path.insert(0, "pose-graph-initialization_amd")
sys.path.insert(0, "tests")
import scene_drivers as SC
g, wave = SC.make_scene("v5000")
SC.write_scene_bulk("/tmp/config45_scene.bin", g, wave, sim_kind=2)
PY
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 PGI_QUIET=1 PGI_DRIVER_REPS=2
for mode in shard waves_guided; do
  rm -rf gpurun_out/${T}_config45_${mode}_trace
  rocprofv3 --kernel-trace --stats -d gpurun_out/${T}_config45_${mode}_trace -o $T -- pose-graph-initialization_amd/test_distributed /tmp/config45_scene.bin /tmp/config45_out $mode > gpurun_out/${T}_config45_${mode}_trace.log 2>&1
  python3 scripts/rocpd_summary.py $(find gpurun_out/${T}_config45_${mode}_trace -name "*.db" | head -1) > gpurun_out/${T}_config45_${mode}_trace_summary.txt 2>&1
  grep "seconds:" gpurun_out/${T}_config45_${mode}_trace.log | cut -c1-200
  head -16 gpurun_out/${T}_config45_${mode}_trace_summary.txt | cut -c1-150
done
rm -f /tmp/config45_scene.bin /tmp/config45_out*
