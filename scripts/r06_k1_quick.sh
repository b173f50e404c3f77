# K1 quick check: parity, config 2's shape at each wavefront count, the dense V = 5000 rows.  Usage: bash scripts/r06_k1_quick.sh [tag]
T=${1:-r06}
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/$T
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/$T/parity.txt 2>&1; tail -2 gpurun_out/$T/parity.txt
timeout 600 python scripts/k1_nw_uniform.py 2000 5000,10000,20000 "${UNI:-nw=4 nw=2 nw=1}" 2>&1 | grep -v amdgpu.ids > gpurun_out/$T/uniform.txt; cat gpurun_out/$T/uniform.txt
K1D_ROUNDS=4 timeout 900 python scripts/k1_dense_bench.py ${DENSE:-nw=1 nw=2 nw=4} 2>&1 | grep "^nw\|^lib" > gpurun_out/$T/dense.txt; cat gpurun_out/$T/dense.txt
