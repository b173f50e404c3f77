set -x
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out
cd $R
python3 scripts/match_bench.py > gpurun_out/match_bench.log 2>&1
rm -rf gpurun_out/match_trace
rocprofv3 --kernel-trace --stats -d gpurun_out/match_trace -o r01 -- python3 scripts/match_bench.py > gpurun_out/match_trace.log 2>&1
python3 scripts/rocpd_summary.py $(find gpurun_out/match_trace -name '*.db') > gpurun_out/match_summary.txt 2>&1
cat gpurun_out/match_bench.log gpurun_out/match_summary.txt
