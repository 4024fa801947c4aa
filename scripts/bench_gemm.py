"""Micro-benchmark of tal_linear_fwd on the TDS shapes of a 1-hour clip (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import ops

dev = torch.device("cuda:0")
shapes = [(179991, 800), (89986, 1120), (44983, 1440)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for M, C in shapes:
    x = torch.randn(M, C, device=dev)
    w = torch.randn(C, C, device=dev) / C ** 0.5
    b = torch.randn(C, device=dev)
    res = torch.randn(M, C, device=dev)
    y = torch.empty(M, C, device=dev)
    for mode in ((2, 1, 2, 1) if os.environ.get("MODE_ORDER") else (1, 2)):
        for _ in range(2):
            ops.linear(x, w, b, mode=mode, res=res, alpha=0.3, out=y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 8
        e0.record()
        for _ in range(n):
            ops.linear(x, w, b, mode=mode, res=res, alpha=0.3, out=y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print("M=%6d N=K=%4d mode %d: %.3f ms  %.1f TFLOP/s" % (M, C, mode, ms, 2.0 * M * C * C / ms / 1e9))
