"""Grouped convs on split-format activations vs the fp32-format kernels (1-hour shapes): correctness + time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import ops
dev = torch.device("cuda:0")
G = 80
def unsplit(buf, rows, K):
    h = buf.view(torch.float16).view(rows, K // 32, 2, 32).float()
    return (h[:, :, 0] + h[:, :, 1] / 2048.0).reshape(rows, K)
def timeit(fn, n=6):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
scale = float(os.environ.get("TSCALE", "1"))
for T, cg in ((179991, 10), (89986, 14), (44983, 18)):
    T = int(T * scale)
    C = G * cg
    x = torch.randn(1, T, C, device=dev)
    w = torch.randn(C, cg, 21, device=dev) / (21 * cg) ** 0.5
    b = torch.randn(C, device=dev)
    wf = ops.pack_gconv_f16x3_weight(w, G)
    y = ops.gconv_res_f16x3(x, wf, b, 0.25, G)
    xs = ops.split_f16x3(x.view(T, C))
    ys = ops.gconv_res_split(xs, (1, T, C), wf, b, 0.25, G)
    err = float((unsplit(ys, T, C) - y.view(T, C)).abs().max())
    t0 = timeit(lambda: ops.gconv_res_f16x3(x, wf, b, 0.25, G))
    t1 = timeit(lambda: ops.gconv_res_f16x3(x, wf, b, 0.25, G, want_split=True))
    t2 = timeit(lambda: ops.gconv_res_split(xs, (1, T, C), wf, b, 0.25, G))
    t3 = timeit(lambda: ops.split_f16x3(y.view(T, C)))
    print("gconv_res cg=%d T=%d: fp32->fp32 %.3f ms | fp32->fp32+split %.3f | split->split %.3f | split pass %.3f | max err %.2e"
          % (cg, T, t0, t1, t2, t3, err), flush=True)
for T, cin, cout in ((179991, 10, 14), (89986, 14, 18)):
    T = int(T * scale)
    x = torch.randn(1, T, G * cin, device=dev)
    w = torch.randn(G * cout, cin, 21, device=dev) / (21 * cin) ** 0.5
    b = torch.randn(G * cout, device=dev)
    wf = ops.pack_gconv_f16x3_weight(w, G, stride=2)
    y = ops.gconv_s2_f16x3(x, wf, b, G * cout, G)
    To = y.shape[1]
    xs = ops.split_f16x3(x.view(T, G * cin))
    ys = ops.gconv_s2_split(xs, (1, T, G * cin), True, wf, b, G * cout, G)
    ys2 = ops.gconv_s2_split(x, (1, T, G * cin), False, wf, b, G * cout, G)
    e1 = float((unsplit(ys, To, G * cout) - y.view(To, -1)).abs().max())
    e2 = float((unsplit(ys2, To, G * cout) - y.view(To, -1)).abs().max())
    t0 = timeit(lambda: ops.gconv_s2_f16x3(x, wf, b, G * cout, G))
    t2 = timeit(lambda: ops.gconv_s2_split(xs, (1, T, G * cin), True, wf, b, G * cout, G))
    t3 = timeit(lambda: ops.gconv_s2_split(x, (1, T, G * cin), False, wf, b, G * cout, G))
    print("gconv_s2 %d->%d T=%d: fp32->fp32 %.3f ms | split->split %.3f | fp32->split %.3f | err split-in %.2e fp32-in %.2e" % (cin, cout, T, t0, t2, t3, e1, e2), flush=True)
