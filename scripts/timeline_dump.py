#!/usr/bin/env python3
"""Timeline of kernels and memory copies from a rocprofv3 (rocpd sqlite) trace: one line per event of the LAST
`window_ms` of activity, times relative to the window start.  usage: timeline_dump.py <results.db> [window_ms]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 12e6
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
ev = []
if "kernels" in tables:
    cols = [r[1] for r in cur.execute("pragma table_info('kernels')")]
    ev += [("K", n, s, e, q) for n, s, e, q in cur.execute("select name, start, end, %s from kernels" % ("queue_id" if "queue_id" in cols else "0"))]
for t in ("memory_copies", "memory_copy"):
    if t in tables:
        cols = [r[1] for r in cur.execute("pragma table_info('%s')" % t)]
        sz = "size" if "size" in cols else "0"
        nm = "name" if "name" in cols else "'copy'"
        ev += [("C", "%s %d B" % (n, b), s, e, 0) for n, s, e, b in cur.execute("select %s, start, end, %s from %s" % (nm, sz, t))]
        break
ev.sort(key=lambda r: r[2])
if not ev:
    print("no events; tables:", tables); sys.exit(0)
t_end = max(e[3] for e in ev)
t0 = t_end - win
for kind, name, s, e, q in ev:
    if e < t0: continue
    print("%s %9.3f -> %9.3f ms (%7.3f) q%-3s %s" % (kind, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, name[:70]))
