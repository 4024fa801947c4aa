"""Can the MFMA-bound dense layer and the VALU-bound grouped conv share the chip?  Both kernels on
independent buffers: back to back on one stream vs concurrently on two streams."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import ops
dev = torch.device("cuda:0")
T, cg, G = 44983, 18, 80
C = G * cg
x = torch.randn(T, C, device=dev); w = torch.randn(C, C, device=dev) / C ** 0.5; b = torch.randn(C, device=dev)
res = torch.randn(T, C, device=dev); y = torch.empty(T, C, device=dev)
xg = torch.randn(1, T, C, device=dev); wg = torch.randn(C, cg, 21, device=dev) / (21 * cg) ** 0.5
wp = ops.pack_gconv_weight(wg, G)
NG, NC = 8, 20
def gemms():
    for _ in range(NG): ops.linear(x, w, b, mode=2, res=res, alpha=0.3, out=y)
def convs():
    for _ in range(NC): ops.gconv_res(xg, wp, b, 0.25, G)
def wall(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
tg, tc = wall(gemms), wall(convs)
print("serial: %d dense layers %.2f ms, %d grouped convs %.2f ms, sum %.2f ms" % (NG, tg, NC, tc, tg + tc))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): gemms()
    with torch.cuda.stream(s2): convs()
    cur.wait_stream(s1); cur.wait_stream(s2)
print("two streams: %.2f ms" % wall(both))
