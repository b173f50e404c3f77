#!/usr/bin/env python3
"""Summarises rocprofv3 (rocpd sqlite) outputs: per-kernel stats and PMC counter sums.
usage: rocpd_summary.py <results.db> [...]"""
import sqlite3
import sys


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        cur = db.cursor()
        print("==", path)
        print("%-60s %6s %12s %12s %7s  vgpr* agpr sgpr lds   (* rocpd field as recorded, not the code-object count: see profiles/r02_kernel_resources.txt)" % ("kernel", "calls", "total_us", "avg_us", "%"))
        rows = cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()  # view is in us
        meta = {r[0]: r[1:] for r in cur.execute(
            "select name, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size) from kernels group by name")}
        for name, calls, tot, avg, pct in rows:
            m = meta.get(name, ("", "", "", ""))
            print("%-60s %6d %12.1f %12.1f %7.2f  %s %s %s %s" % (name[:60], calls, tot, avg, pct, *m))
        try:
            cols = [r[1] for r in cur.execute("pragma table_info('counters_collection')")]
            if "counter_name" in cols:
                q = ("select kernel_name, counter_name, count(*), sum(value) from counters_collection "
                     "group by kernel_name, counter_name")
                got = cur.execute(q).fetchall()
                if got:
                    print("-- PMC (sum over dispatches; 'n' = dispatch x dimension samples)")
                    for k, c, n, v in got:
                        print("%-60s %-24s n=%-6d sum=%.6g" % (k[:60], c, n, v))
        except sqlite3.Error as e:
            print("no counters:", e)


if __name__ == "__main__":
    main()
