set -x
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out
cd $R
python bench.py --steps 3 --warmup 1 > gpurun_out/bench2.json 2> gpurun_out/bench2.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_trace -o r01 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/prof_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_fetch -o r01 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_write -o r01 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/prof_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d gpurun_out/prof_sq -o r01 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/prof_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d gpurun_out/prof_sq2 -o r01 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/prof_sq2.log 2>&1
find gpurun_out -name "*.csv" | head -30
ls -la gpurun_out/prof_trace/* | head
