import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("value","ms_per_step","roofline","roofline_compute","cpu_baseline","h2d_inclusive","rotation_averaging","rotation_averaging_sequence_graph","tracklets","score_pose_k2","quality"):
    print(k, d.get(k))
print(d["match_descriptors"]["roofline"], d["match_descriptors"]["screened"])
