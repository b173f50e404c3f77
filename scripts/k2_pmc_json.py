#!/usr/bin/env python3
"""profiles/<tag>_k2_pmc.json from the summaries of scripts/profile_k2.sh <tag> (usage: k2_pmc_json.py r04 [git head]):
HBM-side traffic of score_pose_kernel (FETCH_SIZE with the gfx950 x2 correction + WRITE_SIZE) per launch, its trace
average, and the hash of the kernel sources that were profiled (bench.py replays the traffic only while it matches)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"
P, N = 10000, 2000


def read(tag, kernel="score_pose_kernel"):
    vals, avg_us, calls = {}, None, None
    for line in open(os.path.join(OUT, "%s_k2_%s_summary.txt" % (TAG, tag))):
        if kernel not in line or "f64" in line:
            continue
        m = re.search(r"(\S+)\s+n=(\d+)\s+sum=(\S+)", line)
        if m:
            vals[m.group(1)] = float(m.group(3)) / int(m.group(2))
        else:  # "<name (cut at 60 chars)>  calls  total_us  avg_us  %  vgpr agpr sgpr lds"
            f = line.split()
            calls, avg_us = int(f[-8]), float(f[-6])
    return vals, avg_us, calls


def main():
    _, avg_us, calls = read("trace")
    fetch = read("fetch")[0]["FETCH_SIZE"]
    write = read("write")[0]["WRITE_SIZE"]
    hbm = fetch * 1024 * 2 + write * 1024
    alg = P * (17 * N + 72 + 8 + 4 + 8)
    out = {"kernel": "pgi::score_pose_kernel", "pairs": P, "corrs": N, "profile_tag": TAG,
           "workload": "10 000 pairs x 2 000 rows, one model per pair, masks written (scripts/k2_bench.py, scripts/profile_k2.sh)",
           "kernel_us_trace_avg": avg_us, "dispatches": calls,
           "fetch_size_kb_per_launch": round(fetch), "write_size_kb_per_launch": round(write),
           "correction": "gfx950: FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported",
           "hbm_bytes_per_launch": round(hbm), "algorithmic_bytes_per_launch": alg,
           "achieved_GBs_algorithmic": round(alg / (avg_us * 1e-6) / 1e9), "achieved_GBs_measured_traffic": round(hbm / (avg_us * 1e-6) / 1e9),
           "frac_of_8TBs_peak": round(alg / (avg_us * 1e-6) / 8e12, 3),
           "source_sha256": open(os.path.join(OUT, "%s_k2_source_sha256.txt" % TAG)).read().strip(),
           "git_head": sys.argv[2] if len(sys.argv) > 2 else None}
    json.dump(out, open(os.path.join(ROOT, "profiles", "%s_k2_pmc.json" % TAG), "w"), indent=2)
    print(json.dumps(out, indent=2))


if __name__ == "__main__":
    main()
