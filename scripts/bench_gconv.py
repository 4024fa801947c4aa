"""Micro-benchmark of the grouped conv kernels on the TDS shapes of a 1-hour clip (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import ops

dev = torch.device("cuda:0")
G = 80
def timeit(fn, n=6):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for T, cg in ((179991, 10), (89986, 14), (44983, 18)):
    C = G * cg
    x = torch.randn(1, T, C, device=dev)
    w = torch.randn(C, cg, 21, device=dev) / (21 * cg) ** 0.5
    b = torch.randn(C, device=dev)
    wp = ops.pack_gconv_weight(w, G)
    ms = timeit(lambda: ops.gconv_res(x, wp, b, 0.25, G))
    print("gconv_res T=%6d C=%4d: %.3f ms  %.1f TFLOP/s  (%.0f GB/s algorithmic)" % (T, C, ms, 2.0 * T * C * cg * 21 / ms / 1e9, 2 * T * C * 4 / ms / 1e6))
    wf = ops.pack_gconv_f16x3_weight(w, G)
    for split in (False, True):
        ms = timeit(lambda: ops.gconv_res_f16x3(x, wf, b, 0.25, G, want_split=split))
        print("gconv_res_f16x3 (matrix cores%s) T=%6d C=%4d: %.3f ms  %.1f TFLOP/s fp32-equivalent" % (", + split output" if split else "", T, C, ms, 2.0 * T * C * cg * 21 / ms / 1e9))
    x0 = torch.zeros_like(x)
    ms = timeit(lambda: ops.gconv_res_f16x3(x0, wf, b, 0.25, G, want_split=True))
    print("   ... on all-zero input: %.3f ms" % ms)
for T, cin, cout in ((360001, 1, 10), (179991, 10, 14), (89986, 14, 18)):
    x = torch.randn(1, T, G * cin, device=dev)
    w = torch.randn(G * cout, cin, 21, device=dev)
    b = torch.randn(G * cout, device=dev)
    wp = ops.pack_gconv_weight(w, G)
    ms = timeit(lambda: ops.gconv_s2(x, wp, b, G * cout, G))
    To = (T - 21) // 2 + 1
    print("gconv_s2  T=%6d %d->%d: %.3f ms  %.1f TFLOP/s" % (T, G * cin, G * cout, ms, 2.0 * To * G * cout * cin * 21 / ms / 1e9))
    wf = ops.pack_gconv_f16x3_weight(w, G, stride=2)
    if wf is not None:
        ms = timeit(lambda: ops.gconv_s2_f16x3(x, wf, b, G * cout, G))
        print("gconv_s2_f16x3 (matrix cores) T=%6d %d->%d: %.3f ms  %.1f TFLOP/s fp32-equivalent" % (T, G * cin, G * cout, ms, 2.0 * To * G * cout * cin * 21 / ms / 1e9))
