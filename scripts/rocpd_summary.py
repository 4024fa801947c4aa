#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel stats table that
`rocprofv3 --kernel-trace --stats` reports (name, calls, total/avg/min/max duration, %).
usage: scripts/rocpd_summary.py results.db [> profiles/xxx_kernel_stats.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                  "from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
total = sum(r[2] for r in rows) or 1
print("%-88s %7s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
for n, c, s, a, mn, mx in rows:
    n = n if len(n) <= 88 else n[:85] + "..."
    print("%-88s %7d %14d %12.0f %12d %12d %6.2f%%" % (n, c, s, a, mn, mx, 100.0 * s / total))
