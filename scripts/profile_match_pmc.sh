set -x
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out
cd $R
rm -rf gpurun_out/match_pmc1 gpurun_out/match_pmc2
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d gpurun_out/match_pmc1 -o r01 -- python3 scripts/match_bench.py > gpurun_out/match_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d gpurun_out/match_pmc2 -o r01 -- python3 scripts/match_bench.py > gpurun_out/match_pmc2.log 2>&1
python3 scripts/rocpd_summary.py $(find gpurun_out/match_pmc1 gpurun_out/match_pmc2 -name '*.db') 2>&1 | grep -v "at::native\|rocclr" > gpurun_out/match_pmc_summary.txt
cat gpurun_out/match_pmc_summary.txt; tail -3 gpurun_out/match_pmc1.log
