"""The 18-channel TDSBlock conv on split-form activations: time-shift-packed kernel against the two-M-tile kernel (option
gconv_no_shift18), 1-hour shape and the 8-segment shape, interleaved A / B with 20 launches per sample."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import ops, _native as N
dev = torch.device("cuda:0")
G, cg = 80, 18
C = G * cg
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for B, T in ((1, 44983), (8, 3733), (1, 3733), (1, 376)):
    x = torch.randn(B, T, C, device=dev)
    w = torch.randn(C, cg, 21, device=dev) / (21 * cg) ** 0.5
    b = torch.randn(C, device=dev)
    wf = ops.pack_gconv_f16x3_weight(w, G)
    xs = ops.split_f16x3(x.view(B * T, C))
    res = {0: [], 1: []}
    for rep in range(3):
        for no in (0, 1):
            N.set_option("gconv_no_shift18", no)
            res[no].append(timeit(lambda: ops.gconv_res_split(xs, (B, T, C), wf, b, 0.25, G)))
    N.set_option("gconv_no_shift18", 0)
    print("B=%d T=%6d: shift-packed %.4f ms (%s) | two M tiles %.4f ms (%s)" % (B, T, min(res[0]), " ".join("%.4f" % v for v in res[0]), min(res[1]), " ".join("%.4f" % v for v in res[1])), flush=True)
