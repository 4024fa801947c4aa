import os, sys
sys.path.insert(0, "/root/repo")
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import ops
dev = torch.device("cuda:0")
def t(M, N, K, n=200):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev); y = torch.empty(M, N, device=dev)
    for _ in range(20): ops.linear(x, w, b, mode=2, res=res, alpha=0.3, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.linear(x, w, b, mode=2, res=res, alpha=0.3, out=y)
    e1.record(); torch.cuda.synchronize()
    print("M=%d N=%d K=%d: %.2f us" % (M, N, K, e0.elapsed_time(e1) / n * 1e3))
for K in (512, 544, 1024, 1056, 2048, 2080, 4096, 4128):
    t(64, 512, K)
t(64, 2048, 512); t(64, 2048, 544)
