# Does binding the host team to the GPU's NUMA node matter?  Interleaved processes, the whole process pinned to the near / far
# socket (taskset) or left alone, with and without the team's own binding (PGI_HOST_NUMA=0 switches it off).  GPU box.
lscpu | grep -i -E "numa node" | head -4
for d in /sys/class/drm/card0/device; do echo "$d: numa_node $(cat $d/numa_node 2>/dev/null)"; done
N0=$(cat /sys/devices/system/node/node0/cpulist); N1=$(cat /sys/devices/system/node/node1/cpulist 2>/dev/null)
run() { PGI_DRIVER_REPS=5 "$@" python scripts/config45_bench.py v5000 waves,shard 2>&1 | grep "seconds:" | sed -e "s/.*mode \([a-z_]*\) .*seconds: [a-zA-Z+*,() ]* \([0-9.]*\),.*/\1 \2/" | awk '{a[$1]=a[$1]" "$2} END {for (m in a) printf "%s %s | ", m, a[m]; print ""}'; }
for round in 1 2 3; do
  for bind in 1 0; do
    export PGI_HOST_NUMA=$bind
    echo "team binding $bind, process free:    $(run env)"
    [ -n "$N1" ] && echo "team binding $bind, process far:     $(run taskset -c $N1)"
    echo "team binding $bind, process near:    $(run taskset -c $N0)"
  done
done
