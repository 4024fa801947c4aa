"""Can the split pass (pure HBM traffic) hide behind the matrix-core grouped conv (SIMD-issue bound)?  Both kernels on
independent buffers: back to back on one stream vs concurrently on two streams, per TDS stage."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import ops
dev = torch.device("cuda:0")
G = 80
def wall(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
for T, cg in ((179991, 10), (89986, 14), (44983, 18)):
    C = G * cg
    xg = torch.randn(1, T, C, device=dev); wg = torch.randn(C, cg, 21, device=dev) / (21 * cg) ** 0.5; b = torch.randn(C, device=dev)
    wf = ops.pack_gconv_f16x3_weight(wg, G)
    xs = torch.randn(T, C, device=dev)
    N = 8
    def convs():
        for _ in range(N): ops.gconv_res_f16x3(xg, wf, b, 0.25, G)
    def splits():
        for _ in range(N): ops.split_f16x3(xs)
    tc, ts = wall(convs), wall(splits)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def both():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1): convs()
        with torch.cuda.stream(s2): splits()
        cur.wait_stream(s1); cur.wait_stream(s2)
    tb = wall(both)
    print("C=%4d: %d grouped convs %.2f ms + %d split passes %.2f ms = %.2f ms serial; two streams %.2f ms" % (C, N, tc, N, ts, tc + ts, tb))
