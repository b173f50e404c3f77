# K2 (score_pose_kernel) alone on config 2's shape: kernel trace + FETCH_SIZE / WRITE_SIZE passes (separate runs; the
# program directly after `--`); scripts/k2_pmc_json.py <tag> writes profiles/<tag>_k2_pmc.json.
# Usage (GPU box): bash scripts/profile_k2.sh r04
T=${1:-r04}
set -x
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out
cd $R
python3 -c "import sys; sys.path.insert(0, 'pose-graph-initialization_amd'); from pyposegraphbuilder import _lib as L; print(L.kernel_source_sha256())" > gpurun_out/${T}_k2_source_sha256.txt
python3 scripts/k2_bench.py > gpurun_out/${T}_k2_bench.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/${T}_k2_trace -o $T -- python3 scripts/k2_bench.py > gpurun_out/${T}_k2_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${T}_k2_fetch -o $T -- python3 scripts/k2_bench.py > gpurun_out/${T}_k2_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${T}_k2_write -o $T -- python3 scripts/k2_bench.py > gpurun_out/${T}_k2_write.log 2>&1
for d in trace fetch write; do
  python3 scripts/rocpd_summary.py $(find gpurun_out/${T}_k2_$d -name "*.db" | head -1) > gpurun_out/${T}_k2_${d}_summary.txt 2>&1
  rm -rf gpurun_out/${T}_k2_$d   # (tens of MB each; gpurun brings back at most 64 MiB)
done
cat gpurun_out/${T}_k2_bench.log; cat gpurun_out/${T}_k2_*_summary.txt | grep -v "^==" | grep -i "score_pose\|kernel " | cut -c1-200
