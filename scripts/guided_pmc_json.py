#!/usr/bin/env python3
"""gpurun_out/<tag>_guided_pmc_summaries.txt (scripts/profile_guided_pmc.sh <tag>) -> profiles/<tag>_guided_pmc.json: the counters of
the guided scan kernel (guided_scan_flat_kernel since r05; GUIDED_KERNEL=... for another) on config 3 from features, per launch, tied to the sha256 of csrc/pgi_match.hip (bench.py replays them
in config3_from_features.dominant_kernel only while that hash matches).  usage: guided_pmc_json.py r05 [git head]"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pose-graph-initialization_amd"))
from pyposegraphbuilder import _lib as L
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"
K = os.environ.get("GUIDED_KERNEL", "guided_sum_kernel<24>")   # the dominant kernel of the scan; its companions are added below
COMPANIONS = () if "GUIDED_KERNEL" in os.environ else ("guided_deal_kernel<24>", "guided_redo_kernel<20>")
c, avg_us, calls, pct = {}, None, None, None
comp = {k: {} for k in COMPANIONS}
for line in open(os.path.join(ROOT, "gpurun_out", "%s_guided_pmc_summaries.txt" % TAG)):
    who = K if K in line else next((k for k in COMPANIONS if k in line), None)
    if who is None:
        continue
    m = re.search(r"(\S+)\s+n=(\d+)\s+sum=(\S+)", line)
    if m:
        (c if who == K else comp[who])[m.group(1)] = float(m.group(3)) / int(m.group(2))   # per launch
    else:
        nums = [x for x in line.split() if re.fullmatch(r"[0-9.]+", x)]
        if who == K:
            calls, avg_us, pct = int(nums[0]), float(nums[2]), float(nums[3])
        else:
            comp[who]["avg_us"] = float(nums[2])
fetch_b, write_b = c["FETCH_SIZE"] * 1024 * 2, c["WRITE_SIZE"] * 1024     # gfx950: FETCH_SIZE tallies 64 B per 128-B request
scan_us = avg_us + sum(v.get("avg_us", 0.0) for v in comp.values())
scan_fetch = fetch_b + sum(v.get("FETCH_SIZE", 0.0) * 2048 for v in comp.values())
scan_write = write_b + sum(v.get("WRITE_SIZE", 0.0) * 1024 for v in comp.values())
# algorithmic bytes per launch (a wave of image pairs): both descriptor sets (512 B per keypoint) and the 48-byte records once per pair
FILES = ["csrc/pgi_match.hip"]
out = {"kernel": K.replace(", fals", ", false>") + " (+ guided_deal_kernel, guided_redo_kernel; config 3 from features: 340 views x 8000 keypoints, waves of 512 pairs, mode 4)",
       "profile_tag": TAG, "source_files": FILES, "source_sha256": L.kernel_source_sha256(tuple(FILES)),
       "git_head": sys.argv[2] if len(sys.argv) > 2 else None,
       "kernel_us_trace_avg": avg_us, "dispatches": calls, "share_of_gpu_time": round(pct / 100.0, 4),
       "scan_us_all_kernels": round(scan_us, 1),
       "companion_kernels": {k: {"avg_us": v.get("avg_us"), "hbm_bytes_per_launch": int(v.get("FETCH_SIZE", 0.0) * 2048 + v.get("WRITE_SIZE", 0.0) * 1024)}
                             for k, v in comp.items()},
       "hbm_bytes_per_launch": int(scan_fetch + scan_write), "fetch_bytes_per_launch": int(scan_fetch), "write_bytes_per_launch": int(scan_write),
       "algorithmic_bytes_per_launch": None, "traffic_over_algorithmic": None,
       # (priced below from the ALGORITHMIC bytes; the counter traffic over the same time is kept beside it)
       "achieved_GBs": None, "frac_hbm": None,
       "traffic_GBs": round((scan_fetch + scan_write) / (scan_us * 1e-6) / 1e9, 1), "traffic_frac_hbm": round((scan_fetch + scan_write) / (scan_us * 1e-6) / 8e12, 4),
       "valu_issue_busy_frac": round(c["SQ_INSTS_VALU"] * 4 / (avg_us * 1e-6 * 2.4e9 * 1024), 3) if "SQ_INSTS_VALU" in c else None,
       "wave_waiting_frac": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3) if "SQ_WAVE_CYCLES" in c else None,
       "lane_utilisation": None, "vgprs": 256 if "tile" in K else (252 if "sum" in K else 249), "spilled_vgprs": 4 if "tile" in K else 0,
       "correction": "FETCH_SIZE x2 on gfx950 (64 B tallied per 128-B request); WRITE_SIZE as reported",
       "counters_per_launch": {k: v for k, v in sorted(c.items())}}
# ~506 image pairs per launch on this workload: 2 x 8000 descriptors x 512 B + 8000 x 48 B records per pair
pairs_per_launch = 6078.0 / max(calls or 12, 1) * (2 if (calls or 12) > 12 else 1)
alg = pairs_per_launch * (2 * 8000 * 512 + 8000 * 48)
out["algorithmic_bytes_per_launch"] = int(alg)
out["achieved_GBs"] = round(alg / (scan_us * 1e-6) / 1e9, 1)          # algorithmic bytes / time of all kernels of the scan
out["frac_hbm"] = round(alg / (scan_us * 1e-6) / 8e12, 4)              # (r05 divided the COUNTER traffic, 0.287 instead of 0.204)
out["traffic_over_algorithmic"] = round((scan_fetch + scan_write) / alg, 2)   # all kernels of the scan, the lists they hand over included
if "SQ_INSTS_VALU_FMA_F64" in c and "SQ_ACTIVE_INST_VALU" in c:
    # useful lane-FMAs per launch: 6.3e9 (round 3: 49 M candidate sums x 128 elements, the same work for every variant of the scan)
    out["note_lanes"] = "lane utilisation of the summation loop = 6.3e9 useful lane-FMAs per launch / (64 x SQ_INSTS_VALU_FMA_F64): a lower bound, the gate's few f64 FMAs are in the count"
    out["lane_utilisation"] = round(6.3e9 / (64.0 * c["SQ_INSTS_VALU_FMA_F64"]), 2)
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_guided_pmc.json" % TAG), "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
