set -x
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out
cd $R
rm -rf gpurun_out/prof_trace
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_trace -o r01 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/prof_trace.log 2>&1
python3 scripts/rocpd_summary.py $(find gpurun_out/prof_trace -name "*.db") > gpurun_out/k1_trace_summary.txt 2>&1
cat gpurun_out/k1_trace_summary.txt; tail -1 gpurun_out/prof_trace.log | cut -c1-400
