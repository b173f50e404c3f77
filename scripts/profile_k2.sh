set -x
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out
cd $R
python3 scripts/k2_bench.py > gpurun_out/k2_bench.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/k2_trace -o r01 -- python3 scripts/k2_bench.py > gpurun_out/k2_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/k2_fetch -o r01 -- python3 scripts/k2_bench.py > gpurun_out/k2_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/k2_write -o r01 -- python3 scripts/k2_bench.py > gpurun_out/k2_write.log 2>&1
cat gpurun_out/k2_bench.log
