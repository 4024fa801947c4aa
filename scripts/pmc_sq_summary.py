#!/usr/bin/env python3
"""Per-kernel SQ counter means of a rocprofv3 --pmc pass over bench.py, with the derived matrix-pipe utilisation:
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)      (both in shader cycles)
  clock     = GRBM_GUI_ACTIVE / 8 / kernel duration                                    (effective GHz under the profiler)
usage: scripts/pmc_sq_summary.py results.db [kernel-substring]"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
pat = "%" + (sys.argv[2] if len(sys.argv) > 2 else "tal") + "%"
rows = db.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                  "where kernel_name like ? group by 1, 2", (pat,)).fetchall()
k = defaultdict(dict)
for name, ctr, n, val, dur in rows:
    k[name][ctr] = val
    k[name]["_n"] = n
    k[name]["_ns"] = dur
print("%-70s %6s %10s %9s %9s %9s %9s" % ("kernel", "calls", "avg_us", "GHz", "mfma_busy", "wait_any", "wait_lds"))
for name, c in sorted(k.items(), key=lambda kv: -kv[1]["_ns"] * kv[1]["_n"]):
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if gui <= 0:
        continue
    wave = c.get("SQ_WAVE_CYCLES", 0.0)
    print("%-70s %6d %10.1f %9.2f %8.1f%% %8.1f%% %8.1f%%" % (
        name[:70], c["_n"], c["_ns"] / 1e3, gui / c["_ns"], 100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * gui),
        100.0 * c.get("SQ_WAIT_ANY", 0.0) / wave if wave else 0.0, 100.0 * c.get("SQ_WAIT_INST_LDS", 0.0) / wave if wave else 0.0))
