#!/usr/bin/env python3
"""profiles/rN_pmc_traffic.json from the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py:
call-weighted mean fabric bytes per dense-layer launch, corrected as MI355X_MICROARCH.md prescribes for gfx950
(FETCH_SIZE x 2 for wide coalesced reads, counter unit KB; WRITE_SIZE as reported).
usage: scripts/pmc_traffic_json.py fetch.db write.db > profiles/r3_pmc_traffic.json"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import dense_kernel_sources_sha256  # noqa: E402  (bench.py quotes this file only while the hash matches its tree)

DENSE = ("gemm_glds_kernel", "gemm_w64_kernel", "head_argmax_kernel")


def per_launch(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)).fetchall()
    vals = [v for n, v in rows if any(d in n for d in DENSE)]
    return len(vals), (sum(vals) / len(vals) if vals else 0.0)


nf, fetch_kb = per_launch(sys.argv[1], "FETCH_SIZE")
nw, write_kb = per_launch(sys.argv[2], "WRITE_SIZE")
print(json.dumps({
    "kernel": "dense-layer launches: tal::gemm_w64_kernel / gemm_glds_kernel (fp16x3 pointwise layers + the fp32 1440->128 layer) + tal::head_argmax_kernel",
    "launches_fetch_pass": nf, "launches_write_pass": nw,
    "FETCH_SIZE_KB_per_launch_raw": fetch_kb, "WRITE_SIZE_KB_per_launch_raw": write_kb, "fetch_correction": 2.0,
    "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
    "kernel_sources_sha256": dense_kernel_sources_sha256(),
    "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python3 bench.py "
            "--no-cpu-baseline --no-exact-pass --steps 2 --warmup 1`; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced "
            "reads on gfx950 (Infinity-Cache hits are counted, so this is fabric traffic, an upper bound on HBM bytes); "
            "WRITE_SIZE uncorrected.  Call-weighted mean over the dense-layer launches; per-kernel values in "
            "rN_pmc_traffic_all_kernels.txt (scripts/pmc_traffic_json.py)."}, indent=1))
