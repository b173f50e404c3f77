# PMC passes for the guided scan kernel on config 3 from features (each pass its own run, program directly after `--`).
# Usage (GPU box): bash scripts/profile_guided_pmc.sh  -> gpurun_out/${T}_guided_pmc_summaries.txt
T=${1:-r05}
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
: > gpurun_out/${T}_guided_pmc_summaries.txt
run() {  # name, counters...
  n=$1; shift
  rm -rf gpurun_out/${T}_guided_$n
  rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/${T}_guided_$n -o $T -- pose-graph-initialization_amd/test_pipeline /tmp/config3_features.bin /tmp/config3_features.out 4 > gpurun_out/${T}_guided_$n.log 2>&1
  python3 scripts/rocpd_summary.py $(find gpurun_out/${T}_guided_$n -name "*.db" | head -1) 2>&1 | grep "guided_\|^kernel" | cut -c1-160 >> gpurun_out/${T}_guided_pmc_summaries.txt
  rm -rf gpurun_out/${T}_guided_$n   # (the traces are tens of MB each; gpurun brings back at most 64 MiB)
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS
run mix SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS
run lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES
cat gpurun_out/${T}_guided_pmc_summaries.txt
rm -f /tmp/config3_features.bin /tmp/config3_features.out
