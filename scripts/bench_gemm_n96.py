"""Mid-sized fp16x3 launches (N = K = 1440): 128 x 96 tiles against 128 x 160 tiles (option gemm_no_n96), bit-identity of the two,
time per layer pair.  python scripts/bench_gemm_n96.py [rows ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import ops, _native as N
lib = N.lib()
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
C = 1440
for M in [int(a) for a in sys.argv[1:]] or [1900, 2500, 3000, 3751, 4200, 5000, 6000, 7200, 7501]:
    gen = torch.Generator().manual_seed(M)
    x = torch.randn(M, C, generator=gen).to(dev)
    w0 = (torch.randn(C, C, generator=gen) / C ** 0.5).to(dev); b0 = torch.randn(C, generator=gen).to(dev)
    w1 = (torch.randn(C, C, generator=gen) / C ** 0.5).to(dev); b1 = torch.randn(C, generator=gen).to(dev)
    xs, w0s, w1s = ops.split_f16x3(x), ops.split_f16x3(w0), ops.split_f16x3(w1)
    nws = lib.tal_linear_workspace_bytes(M, C, C)
    ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)
    flag = torch.zeros(16, dtype=torch.int32, device=dev)
    hs = torch.zeros(M * C * 4, dtype=torch.uint8, device=dev); ys = torch.zeros(M * C * 4, dtype=torch.uint8, device=dev)
    def pair():
        N.check(lib.tal_linear_f16x3_guarded_fwd(N.ptr(xs), N.ptr(w0s), N.ptr(b0), None, 0, 0.0, 1, M, C, C, N.ptr(hs), 1, N.ptr(flag), N.ptr(ws), nws, N.stream_handle()), "relu")
        N.check(lib.tal_linear_f16x3_guarded_fwd(N.ptr(hs), N.ptr(w1s), N.ptr(b1), N.ptr(xs), 1, 0.3, 2, M, C, C, N.ptr(ys), 1, N.ptr(flag), N.ptr(ws), nws, N.stream_handle()), "res")
    out = {}
    for name, v in (("96-wide allowed", 0), ("160-wide only", 1)):
        N.set_option("gemm_no_n96", v)
        pair(); torch.cuda.synchronize()
        out[name] = (timeit(pair), hs.clone(), ys.clone())
    N.set_option("gemm_no_n96", 0)
    a, b = out["96-wide allowed"], out["160-wide only"]
    print("M=%5d: 96-wide allowed %6.1f us | 160-wide only %6.1f us | outputs bit-identical: %s" % (M, a[0], b[0], bool(torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]))), flush=True)
