"""Micro-benchmark of the diarization head (tal_sd_head_fwd) on the 1-hour shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import ops
dev = torch.device("cuda:0")
M = 44983
x = torch.randn(M, 1440, device=dev)
we = torch.randn(128, 1440, device=dev) / 38; be = torch.randn(128, device=dev)
wl = torch.randn(6008, 128, device=dev) / 11; bl = torch.randn(6008, device=dev)
def timeit(fn, n=6):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("feat only        %.3f ms" % timeit(lambda: ops.sd_head(x, we, be, wl, bl, want_logits=False, want_ids=False)))
print("feat+ids (fused) %.3f ms" % timeit(lambda: ops.sd_head(x, we, be, wl, bl, want_logits=False, want_ids=True)))
print("feat+logits+ids  %.3f ms" % timeit(lambda: ops.sd_head(x, we, be, wl, bl, want_logits=True, want_ids=True)))
f, l, i = ops.sd_head(x, we, be, wl, bl, want_logits=True, want_ids=True)
f2, _, i2 = ops.sd_head(x, we, be, wl, bl, want_logits=False, want_ids=True)
print("ids equal:", bool((i == i2).all()), "torch argmax equal:", bool((l.argmax(-1) == i2).all()))
