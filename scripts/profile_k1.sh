# K1 (estimate_pose_kernel) on BASELINE config 2: kernel trace + PMC passes (each in its own run, never combined with
# other trace domains; the program sits directly after `--`), summarised into gpurun_out/<tag>_k1_*.txt; the sha256 of
# the kernel sources that were profiled is recorded next to them.  scripts/k1_pmc_json.py <tag> turns the sums into
# profiles/<tag>_k1_pmc.json.  Usage (GPU box): bash scripts/profile_k1.sh r04
T=${1:-r04}
set -x
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out
cd $R
python3 -c "import sys; sys.path.insert(0, 'pose-graph-initialization_amd'); from pyposegraphbuilder import _lib as L; print(L.kernel_source_sha256())" > gpurun_out/${T}_k1_source_sha256.txt
ARGS="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-variants"
# the same command WITHOUT the profiler, on this very lease: its HIP-event kernel time is what the profiled averages are read against
python3 $ARGS > gpurun_out/${T}_k1_unprofiled_bench.json 2> gpurun_out/${T}_k1_unprofiled_bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/${T}_k1_trace -o $T -- python3 $ARGS > gpurun_out/${T}_k1_trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 -d gpurun_out/${T}_k1_flops -o $T -- python3 $ARGS > gpurun_out/${T}_k1_flops.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_MFMA SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d gpurun_out/${T}_k1_mix -o $T -- python3 $ARGS > gpurun_out/${T}_k1_mix.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU -d gpurun_out/${T}_k1_wait -o $T -- python3 $ARGS > gpurun_out/${T}_k1_wait.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d gpurun_out/${T}_k1_lds -o $T -- python3 $ARGS > gpurun_out/${T}_k1_lds.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${T}_k1_fetch -o $T -- python3 $ARGS > gpurun_out/${T}_k1_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${T}_k1_write -o $T -- python3 $ARGS > gpurun_out/${T}_k1_write.log 2>&1
for d in trace flops mix wait lds fetch write; do
  python3 scripts/rocpd_summary.py $(find gpurun_out/${T}_k1_$d -name "*.db" | head -1) > gpurun_out/${T}_k1_${d}_summary.txt 2>&1
  rm -rf gpurun_out/${T}_k1_$d   # (tens of MB each; gpurun brings back at most 64 MiB)
done
cat gpurun_out/${T}_k1_*_summary.txt | grep -v "^==" | grep -i "estimate_pose\|bucket\|kernel " | cut -c1-200
