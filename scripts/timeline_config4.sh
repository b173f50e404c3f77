# Timeline (kernels + memory copies) of config 4 on the dense V = 5000 scene: where the estimation stage's time goes between
# the uploads and the K1 launches.  Usage (GPU box): bash scripts/timeline_config4.sh [tag] ; env PGI_K1_STREAMED / PGI_K1_NW pass through
T=${1:-r05}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out
cd $R
python3 - <<'PY'
import sys
sys.path.insert(0, "pose-graph-initialization_amd")
sys.path.insert(0, "tests")
import scene_drivers as SC
g, wave = SC.make_scene("v5000")
SC.write_scene_bulk("/tmp/config45_scene.bin", g, wave, sim_kind=2)
PY
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 PGI_QUIET=1 PGI_DRIVER_REPS=3 PGI_HOST_TIMING=1
rm -rf gpurun_out/${T}_config4_timeline
rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/${T}_config4_timeline -o $T -- pose-graph-initialization_amd/test_distributed /tmp/config45_scene.bin /tmp/config45_out shard > gpurun_out/${T}_config4_timeline.log 2>&1
python3 scripts/timeline_dump.py $(find gpurun_out/${T}_config4_timeline -name "*.db" | head -1) 115 > gpurun_out/${T}_config4_timeline.txt 2>&1
grep "seconds:" gpurun_out/${T}_config4_timeline.log | cut -c1-160
tail -n 22 gpurun_out/${T}_config4_timeline.log | grep estimatePoses
rm -rf gpurun_out/${T}_config4_timeline /tmp/config45_scene.bin /tmp/config45_out*
