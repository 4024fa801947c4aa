#!/usr/bin/env python3
"""Per-kernel VGPR/AGPR/SGPR/scratch/LDS/occupancy table for .hip files (compile-only, no GPU).
usage: scripts/kernel_resources.py tal_asrd_amd/csrc/gemm_f32.hip [...]"""
import re
import subprocess
import sys

for src in sys.argv[1:]:
    p = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c",
                        "-Rpass-analysis=kernel-resource-usage", src, "-o", "/dev/null"],
                       capture_output=True, text=True)
    cur = None
    for line in p.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            if cur:
                print(cur)
            name = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
            cur = name.split("(")[0][:70].ljust(70)
        elif cur and re.match(r"(VGPRs|AGPRs|TotalSGPRs|ScratchSize|Occupancy|LDS Size)", t):
            cur += " | " + t.replace(" [bytes/lane]", "").replace(" [waves/SIMD]", "").replace(" [bytes/block]", "")
    if cur:
        print(cur)
