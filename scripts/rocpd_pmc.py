#!/usr/bin/env python3
"""Per-kernel PMC counter means from a rocprofv3 (rocpd sqlite) --pmc run.
usage: scripts/rocpd_pmc.py results.db [kernel-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = "%" + (sys.argv[2] if len(sys.argv) > 2 else "tal::") + "%"
rows = db.execute("select substr(kernel_name, 1, 60), counter_name, count(*), avg(value), avg(duration) "
                  "from counters_collection where kernel_name like ? group by 1, 2 order by 1, 2", (pat,)).fetchall()
for name, ctr, n, val, dur in rows:
    print("%-60s %-28s n=%-4d mean=%.4g  kernel_ns=%.0f" % (name, ctr, n, val, dur))
