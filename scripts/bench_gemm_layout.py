"""Does the relative placement of the operand buffers matter?  Stage-1 dense layer (M=179991, N=K=800) with
x / res / y carved from one allocation at the TDS driver's spacing vs separate allocations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import ops
dev = torch.device("cuda:0")
M, C = 179991, 800
w = torch.randn(C, C, device=dev) / C ** 0.5
b = torch.randn(C, device=dev)
def run(x, res, y, tag):
    for mode in (2, 1, 2, 1):
        for _ in range(3):
            ops.linear(x, w, b, mode=mode, res=res, alpha=0.3, out=y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            ops.linear(x, w, b, mode=mode, res=res, alpha=0.3, out=y)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 8
        print("%-28s mode %d: %.3f ms %.1f TF" % (tag, mode, ms, 2.0 * M * C * C / ms / 1e9))
nf = M * C
for pad in (0, 64, 1024, 4096 + 64):
    big = torch.randn(3 * (nf + pad) + 16, device=dev)
    bufs = [big[i * (nf + pad): i * (nf + pad) + nf].view(M, C) for i in range(3)]
    run(bufs[0], bufs[1], bufs[2], "carved, pad %d floats" % pad)
run(torch.randn(M, C, device=dev), torch.randn(M, C, device=dev), torch.empty(M, C, device=dev), "separate allocations")
