# K1 (estimate_pose_kernel) on BASELINE config 2: kernel trace + PMC passes (each in its own run, never combined with
# other trace domains), summarised into gpurun_out/r02_k1_*.txt; scripts/k1_pmc_json.py turns the sums into
# profiles/r02_k1_pmc.json.  Usage (GPU box): bash scripts/profile_k1_r02.sh
set -x
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out
cd $R
ARGS="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats -d gpurun_out/r02_k1_trace -o r02 -- python3 $ARGS > gpurun_out/r02_k1_trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 -d gpurun_out/r02_k1_flops -o r02 -- python3 $ARGS > gpurun_out/r02_k1_flops.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_MFMA SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d gpurun_out/r02_k1_mix -o r02 -- python3 $ARGS > gpurun_out/r02_k1_mix.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU -d gpurun_out/r02_k1_wait -o r02 -- python3 $ARGS > gpurun_out/r02_k1_wait.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d gpurun_out/r02_k1_lds -o r02 -- python3 $ARGS > gpurun_out/r02_k1_lds.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r02_k1_fetch -o r02 -- python3 $ARGS > gpurun_out/r02_k1_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r02_k1_write -o r02 -- python3 $ARGS > gpurun_out/r02_k1_write.log 2>&1
for d in trace flops mix wait lds fetch write; do
  python3 scripts/rocpd_summary.py $(find gpurun_out/r02_k1_$d -name "*.db" | head -1) > gpurun_out/r02_k1_${d}_summary.txt 2>&1
done
cat gpurun_out/r02_k1_*_summary.txt | grep -v "^==" | grep -i "estimate_pose\|kernel " | cut -c1-200
