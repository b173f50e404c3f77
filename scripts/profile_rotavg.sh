# rotation averaging (pgi_rotation_average) at V = 340 / 1500 / 5000 under rocprofv3 --kernel-trace --stats: which kernels make up the time.
# Usage (GPU box): bash scripts/profile_rotavg.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out
cd $R
rm -rf gpurun_out/r04_rotavg_trace
rocprofv3 --kernel-trace --stats -d gpurun_out/r04_rotavg_trace -o r04 -- python3 scripts/rotavg_bench.py > gpurun_out/r04_rotavg_trace.log 2>&1
python3 scripts/rocpd_summary.py $(find gpurun_out/r04_rotavg_trace -name "*.db" | head -1) > gpurun_out/r04_rotavg_trace_summary.txt 2>&1
head -24 gpurun_out/r04_rotavg_trace_summary.txt | cut -c1-150; tail -4 gpurun_out/r04_rotavg_trace.log
