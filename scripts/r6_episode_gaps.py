#!/usr/bin/env python3
"""Where a generated token's time goes in a real episode: from a rocprofv3 kernel trace (rocpd sqlite) of scripts/bench_episode.py,
per decode step (embed kernel ... pick kernel) the sum of kernel durations, the gaps between the step's kernels, and the gap from the
previous step's pick to this step's embed (host: poll + control flow + the next call's first launch).
usage: scripts/r6_episode_gaps.py results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
steps, cur = [], None
for name, st, en in rows:
    if "embed_kernel" in name:
        cur = {"t0": st, "kern": 0.0, "n": 0, "last_end": st, "inner_gap": 0.0, "by": {}}
    if cur is None:
        continue
    cur["inner_gap"] += max(0, st - cur["last_end"]) / 1e3
    cur["kern"] += (en - st) / 1e3
    cur["n"] += 1
    cur["last_end"] = en
    key = name.split("(")[0].replace("void tal::", "").replace("tal::", "")[:40]
    cur["by"][key] = cur["by"].get(key, 0.0) + (en - st) / 1e3
    if "lm_pick_kernel" in name:
        cur["t1"] = en
        steps.append(cur)
        cur = None
steps = [s for s in steps if s["n"] >= 20]
half = steps[len(steps) // 2:]          # the second run of the script (warm)
gaps = [(b["t0"] - a["t1"]) / 1e3 for a, b in zip(half, half[1:]) if 0 < b["t0"] - a["t1"] < 200e3]
n = len(half)
print("%d decode steps (second half of the trace)" % n)
print("launches per step: %.1f   kernel time per step %.1f us   gaps inside a step %.1f us   pick -> next embed %.1f us (median %.1f)"
      % (sum(s["n"] for s in half) / n, sum(s["kern"] for s in half) / n, sum(s["inner_gap"] for s in half) / n, sum(gaps) / len(gaps), sorted(gaps)[len(gaps) // 2]))
tot = {}
for s in half:
    for k, v in s["by"].items():
        tot[k] = tot.get(k, 0.0) + v
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print("  %-42s %7.1f us per step" % (k, v / n))
