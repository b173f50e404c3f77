set -x
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out
cd $R
rm -rf gpurun_out/prof_fetch gpurun_out/prof_write
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_fetch -o r01 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_write -o r01 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/prof_write.log 2>&1
python3 scripts/rocpd_summary.py $(find gpurun_out/prof_fetch gpurun_out/prof_write -name "*.db") 2>&1 | grep -v "at::native\|rocclr" > gpurun_out/k1_traffic_summary.txt
cat gpurun_out/k1_traffic_summary.txt
