#!/usr/bin/env python3
"""Per-kernel means of every counter of a rocprofv3 --pmc pass.  usage: scripts/pmc_generic.py results.db [kernel-substring]"""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
pat = "%" + (sys.argv[2] if len(sys.argv) > 2 else "tal") + "%"
rows = db.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                  "where kernel_name like ? group by 1, 2", (pat,)).fetchall()
k = defaultdict(dict)
for name, ctr, n, val, dur in rows:
    k[name][ctr] = val; k[name]["_n"] = n; k[name]["_us"] = dur / 1e3
ctrs = sorted({c for v in k.values() for c in v if not c.startswith("_")})
print("%-64s %5s %8s " % ("kernel", "calls", "avg_us") + " ".join("%14s" % c[-14:] for c in ctrs))
for name, c in sorted(k.items(), key=lambda kv: -kv[1]["_us"] * kv[1]["_n"]):
    print("%-64s %5d %8.1f " % (name[:64], c["_n"], c["_us"]) + " ".join("%14.4g" % c.get(x, float("nan")) for x in ctrs))
