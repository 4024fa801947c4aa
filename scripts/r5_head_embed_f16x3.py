import sys, time
sys.path.insert(0, "/root/repo")
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import ops, _native as N
lib = N.lib()
dev = torch.device("cuda:0")
for M in (44983, 3733, 29864):
    K, Nn = 1440, 128
    x = torch.randn(M, K, device=dev); w = torch.randn(Nn, K, device=dev) / K ** 0.5; b = torch.randn(Nn, device=dev)
    xs, wsp = ops.split_f16x3(x), ops.split_f16x3(w)
    y = torch.empty(M, Nn, device=dev)
    nws = lib.tal_linear_workspace_bytes(M, 160, K)
    ws = torch.empty(max(nws, 1 << 26), dtype=torch.uint8, device=dev)
    def f16():
        N.check(lib.tal_linear_f16x3_fwd(N.ptr(xs), N.ptr(wsp), N.ptr(b), None, 0.0, 0, M, Nn, K, N.ptr(y), 0, N.ptr(ws), ws.numel(), N.stream_handle()), "f16x3")
    def f32():
        return ops.linear(x, w, b)
    try:
        f16()
    except Exception as e:
        print("M=%d: fp16x3 N=128 refused: %s" % (M, e)); continue
    ref = (x.double() @ w.double().t() + b.double())
    e16 = float((y.double() - ref).abs().max()); e32 = float((f32().double() - ref).abs().max())
    def t(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    print("M=%d: fp16x3 %.1f us (max err %.2e) | fp32 %.1f us (max err %.2e)" % (M, t(f16), e16, t(f32), e32), flush=True)
