#!/usr/bin/env python3
"""bench.py's multi-GPU graph leg (configs 4 / 5 at SURVEY 8d's density through `world` child ranks of the C++ driver, every
rank's result compared with the single-process run byte for byte) exercised on the ONE GPU of the test box: the ranks share
the device, so dist::attach picks the host transport (RCCL refuses duplicate devices).  Usage: graph_level_world2.py [world]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pose-graph-initialization_amd"))
import bench
world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
g = bench.graph_level(world)
print(json.dumps(g, indent=1))
ok = all(m.get("identical_to_single_process") for v in g.values() if isinstance(v, dict) for m in v.values() if isinstance(m, dict) and "world" in m)
print("identical to the single-process run on every rank:", ok)
sys.exit(0 if ok else 1)
