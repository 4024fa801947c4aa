"""Separate the dense-layer kernel's per-tile fixed cost (prologue + epilogue) from its per-K-step cost:
whole rounds only (tiles = multiple of 2 x CUs), K swept; time = rounds * (c0 + c1 * K/32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import ops

dev = torch.device("cuda:0")
N = int(os.environ.get("FIT_N", "1440"))
M = 128 * 512 * int(os.environ.get("FIT_ROUNDS_PER_NT", "1"))      # tiles = M/128 * N/160 -> N/160 full rounds
rounds = (M // 128) * (N // 160) / 512.0
for mode in (1, 2):
    pts = []
    for K in (320, 640, 960, 1440, 2880, 5760):
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        res = torch.randn(M, N, device=dev)
        y = torch.empty(M, N, device=dev)
        for _ in range(3):
            ops.linear(x, w, b, mode=mode, res=res, alpha=float(os.environ.get('FIT_ALPHA','0.3')), out=y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            ops.linear(x, w, b, mode=mode, res=res, alpha=float(os.environ.get('FIT_ALPHA','0.3')), out=y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        pts.append((K // 32, ms * 1e3 / rounds))
        print("mode %d M=%d N=%d K=%4d: %.3f ms  %.1f TFLOP/s   %.1f us/round" % (mode, M, N, K, ms, 2.0 * M * N * K / ms / 1e9, ms * 1e3 / rounds))
    a = np.array(pts)
    c1, c0 = np.polyfit(a[:, 0], a[:, 1], 1)
    print("  fit: per-tile fixed %.1f us, per K-step %.3f us (ideal 2 x 80 MFMA x 64 cyc / 2.4 GHz = 4.267 us)" % (c0, c1))
