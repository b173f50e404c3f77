# Exact f32 matcher (desc_top2_kernel, PGI_MATCH_SCREEN=0) on 8 images x 8000 keypoints, 56 pairs: kernel trace + two PMC
# passes (own runs), summarised into gpurun_out/r02_match_exact_*.txt.  Usage (GPU box): bash scripts/profile_match_exact.sh
set -x
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out
cd $R
export PGI_MATCH_SCREEN=0
ARGS="scripts/match_variant_probe.py"
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r02_mx_trace -o r02 -- python3 $ARGS > gpurun_out/r02_mx_trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d gpurun_out/r02_mx_wait -o r02 -- python3 $ARGS > gpurun_out/r02_mx_wait.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d gpurun_out/r02_mx_lds -o r02 -- python3 $ARGS > gpurun_out/r02_mx_lds.log 2>&1
for d in trace wait lds; do
  python3 scripts/rocpd_summary.py $(find gpurun_out/r02_mx_$d -name "*.db" | head -1) > gpurun_out/r02_match_exact_${d}_summary.txt 2>&1
done
cat gpurun_out/r02_match_exact_*_summary.txt | grep -v "^==" | grep -i "top2\|kernel " | cut -c1-260
