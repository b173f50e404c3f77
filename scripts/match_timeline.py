#!/usr/bin/env python3
"""Per-kernel timeline of the last pgi_match_descriptors_batch call in a rocprofv3 --kernel-trace database
(scripts/profile_match.sh writes gpurun_out/match_trace/*.db)."""
import glob, sqlite3, sys
db = sys.argv[1] if len(sys.argv) > 1 else glob.glob("gpurun_out/match_trace/*.db")[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "info_kernel_symbol" in t][0]
rows = c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
last = [i for i, r in enumerate(rows) if "match_select" in r[0]][-1]
first = last
while first > 0 and "match_select" not in rows[first - 1][0] and "prepare" not in rows[first - 1][0]:
    first -= 1
t0 = rows[first][1]
for n, s, e in rows[first:last + 1]:
    print("%9.3f -> %9.3f ms  %8.1f us  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e3, n[:64]))
print("span %.3f ms" % ((rows[last][2] - t0) / 1e6))
