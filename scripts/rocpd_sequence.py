#!/usr/bin/env python3
"""Time-ordered kernel sequence of the last bench step from a rocprofv3 (rocpd sqlite) kernel trace.
usage: scripts/rocpd_sequence.py results.db [n_last]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = rows[-n:]
prev_end = rows[0][1]
for name, st, en in rows:
    print("%-70s gap %8.1f us  dur %9.1f us" % (name[:70], (st - prev_end) / 1e3, (en - st) / 1e3))
    prev_end = en
