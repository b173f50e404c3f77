#!/usr/bin/env python3
"""gpurun_out/<tag>_guided_pmc_summaries.txt (scripts/profile_guided_pmc.sh <tag>) -> profiles/<tag>_guided_pmc.json: the counters of
the guided scan kernel (guided_scan_flat_kernel since r05; GUIDED_KERNEL=... for another) on config 3 from features, per launch, tied to the sha256 of csrc/pgi_match.hip (bench.py replays them
in config3_from_features.dominant_kernel only while that hash matches).  usage: guided_pmc_json.py r05 [git head]"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-graph-initialization_amd"))
from pyposegraphbuilder import _lib as L
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"
K = os.environ.get("GUIDED_KERNEL", "guided_scan_flat_kernel<20, fals")   # (rocpd_summary.py cuts the names at 60 characters)
c, avg_us, calls, pct = {}, None, None, None
for line in open(os.path.join(ROOT, "gpurun_out", "%s_guided_pmc_summaries.txt" % TAG)):
    if K not in line:
        continue
    m = re.search(r"(\S+)\s+n=(\d+)\s+sum=(\S+)", line)
    if m:
        c[m.group(1)] = float(m.group(3)) / int(m.group(2))   # per launch
    else:
        nums = [x for x in line.split() if re.fullmatch(r"[0-9.]+", x)]
        calls, avg_us, pct = int(nums[0]), float(nums[2]), float(nums[3])
fetch_b, write_b = c["FETCH_SIZE"] * 1024 * 2, c["WRITE_SIZE"] * 1024     # gfx950: FETCH_SIZE tallies 64 B per 128-B request
# algorithmic bytes per launch (a wave of image pairs): both descriptor sets (512 B per keypoint) and the 48-byte records once per pair
FILES = ["csrc/pgi_match.hip"]
out = {"kernel": K.replace(", fals", ", false>") + " (config 3 from features: 340 views x 8000 keypoints, waves of 512 pairs, mode 4)",
       "profile_tag": TAG, "source_files": FILES, "source_sha256": L.kernel_source_sha256(tuple(FILES)),
       "git_head": sys.argv[2] if len(sys.argv) > 2 else None,
       "kernel_us_trace_avg": avg_us, "dispatches": calls, "share_of_gpu_time": round(pct / 100.0, 4),
       "hbm_bytes_per_launch": int(fetch_b + write_b), "fetch_bytes_per_launch": int(fetch_b), "write_bytes_per_launch": int(write_b),
       "algorithmic_bytes_per_launch": None, "traffic_over_algorithmic": None,
       "achieved_GBs": round((fetch_b + write_b) / (avg_us * 1e-6) / 1e9, 1), "frac_hbm": round((fetch_b + write_b) / (avg_us * 1e-6) / 8e12, 4),
       "valu_issue_busy_frac": round(c["SQ_INSTS_VALU"] * 4 / (avg_us * 1e-6 * 2.4e9 * 1024), 3) if "SQ_INSTS_VALU" in c else None,
       "wave_waiting_frac": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3) if "SQ_WAVE_CYCLES" in c else None,
       "lane_utilisation": None, "vgprs": 256 if "tile" in K else 249, "spilled_vgprs": 4 if "tile" in K else 0,
       "correction": "FETCH_SIZE x2 on gfx950 (64 B tallied per 128-B request); WRITE_SIZE as reported",
       "counters_per_launch": {k: v for k, v in sorted(c.items())}}
# ~506 image pairs per launch on this workload: 2 x 8000 descriptors x 512 B + 8000 x 48 B records per pair
pairs_per_launch = 6078.0 / max(calls or 12, 1) * (2 if (calls or 12) > 12 else 1)
alg = pairs_per_launch * (2 * 8000 * 512 + 8000 * 48)
out["algorithmic_bytes_per_launch"] = int(alg)
out["traffic_over_algorithmic"] = round((fetch_b + write_b) / alg, 2)
if "SQ_INSTS_VALU_FMA_F64" in c and "SQ_ACTIVE_INST_VALU" in c:
    # useful lane-FMAs per launch: 6.3e9 (round 3: 49 M candidate sums x 128 elements, the same work for every variant of the scan)
    out["note_lanes"] = "lane utilisation of the summation loop = 6.3e9 useful lane-FMAs per launch / (64 x SQ_INSTS_VALU_FMA_F64): a lower bound, the gate's few f64 FMAs are in the count"
    out["lane_utilisation"] = round(6.3e9 / (64.0 * c["SQ_INSTS_VALU_FMA_F64"]), 2)
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_guided_pmc.json" % TAG), "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
