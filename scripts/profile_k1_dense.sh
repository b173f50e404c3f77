# K1 on the rows of the dense V = 5000 scene (BASELINE configs 4/5 at SURVEY 8d's density; one resident call of 106 151 pairs,
# scripts/k1_dense_bench.py): kernel trace with ONE LINE PER SIZE CLASS (the classes are different kernel instances: rows whole
# in LDS / hybrid), the PMC passes of scripts/profile_k1.sh (each in its own run, never combined with other trace domains; the
# program directly after `--`), and the per-phase cycle accounting of the instrumented build on the <= 1344-row pairs.
# Usage (GPU box): bash scripts/profile_k1_dense.sh r05   -> gpurun_out/<tag>_k1_dense_{trace,pmc,phases}.txt
T=${1:-r05}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out
cd $R
export K1D_ROUNDS=3
SHA=$(python3 -c "import sys; sys.path.insert(0, 'pose-graph-initialization_amd'); from pyposegraphbuilder import _lib as L; print(L.kernel_source_sha256())")
ARGS="scripts/k1_dense_bench.py nw=0"
OUT=gpurun_out/${T}_k1_dense
echo "# kernel sources sha256 $SHA" > ${OUT}_trace.txt
python3 $ARGS 2>&1 | grep "^scene\|pairs with\|^nw" >> ${OUT}_trace.txt   # un-profiled, same lease
rocprofv3 --kernel-trace --stats -d ${OUT}_t -o $T -- python3 $ARGS > ${OUT}_t.log 2>&1
python3 scripts/rocpd_summary.py $(find ${OUT}_t -name "*.db" | head -1) 2>&1 | grep "estimate_pose\|bucket\|^kernel" | cut -c1-170 >> ${OUT}_trace.txt
echo "# kernel sources sha256 $SHA" > ${OUT}_pmc.txt
run() {  # name, counters...
  n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d ${OUT}_$n -o $T -- python3 $ARGS > ${OUT}_$n.log 2>&1
  python3 scripts/rocpd_summary.py $(find ${OUT}_$n -name "*.db" | head -1) 2>&1 | grep "estimate_pose" | cut -c1-170 >> ${OUT}_pmc.txt
  rm -rf ${OUT}_$n
}
run flops SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64
run mix SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_MFMA SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES
run wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
rm -rf ${OUT}_t
make -C pose-graph-initialization_amd libpgi_prof.so > gpurun_out/${T}_k1_dense_prof_build.log 2>&1   # (the instrumented build of THESE sources)
echo "# kernel sources sha256 $SHA; instrumented build (libpgi_prof.so), pairs of at most 1344 rows, four wavefronts per pair (PGI_K1_NW=4), then one (PGI_K1_NW=1)" > ${OUT}_phases.txt
PGI_K1_NW=4 python3 scripts/profile_phases.py 20000 v5000:1344 2>&1 | grep -v amdgpu.ids >> ${OUT}_phases.txt
PGI_K1_NW=1 python3 scripts/profile_phases.py 20000 v5000:1344 2>&1 | grep -v amdgpu.ids >> ${OUT}_phases.txt
python3 scripts/profile_phases.py 8192 600 2>&1 | grep -v amdgpu.ids >> ${OUT}_phases.txt
cat ${OUT}_trace.txt; head -30 ${OUT}_pmc.txt
