"""Short-input dense pair (csrc/gemm_s64.hip) on 64-row tiles (4 waves) against 32-row tiles (2 waves per workgroup).
python scripts/bench_gemm_s64_rows.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import ops, _native as N
lib = N.lib()
dev = torch.device("cuda:0")

def timeit(fn, n=40):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

N.set_option("gemm_s64_below", 1 << 20)
for C, rows in ((800, (200, 501, 751, 1501)), (1120, (200, 376, 751, 1501)), (1440, (200, 376, 751, 1501))):
    for M in rows:
        gen = torch.Generator().manual_seed(M + C)
        x = torch.randn(M, C, generator=gen).to(dev)
        w0 = (torch.randn(C, C, generator=gen) / C ** 0.5).to(dev); b0 = torch.randn(C, generator=gen).to(dev)
        w1 = (torch.randn(C, C, generator=gen) / C ** 0.5).to(dev); b1 = torch.randn(C, generator=gen).to(dev)
        xs, w0s, w1s = ops.split_f16x3(x), ops.split_f16x3(w0), ops.split_f16x3(w1)
        nws = lib.tal_linear_workspace_bytes(M, C, C)
        ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)
        flag = torch.zeros(16, dtype=torch.int32, device=dev)
        hs = torch.zeros(M * C * 4, dtype=torch.uint8, device=dev)
        ys = torch.zeros(M * C * 4, dtype=torch.uint8, device=dev)
        def pair():
            N.check(lib.tal_linear_f16x3_guarded_fwd(N.ptr(xs), N.ptr(w0s), N.ptr(b0), None, 0, 0.0, 1, M, C, C, N.ptr(hs), 1, N.ptr(flag),
                                                     N.ptr(ws), nws, N.stream_handle()), "relu layer")
            N.check(lib.tal_linear_f16x3_guarded_fwd(N.ptr(hs), N.ptr(w1s), N.ptr(b1), N.ptr(xs), 1, 0.3, 2, M, C, C, N.ptr(ys), 1, N.ptr(flag),
                                                     N.ptr(ws), nws, N.stream_handle()), "residual layer")
        res = {}
        for rows_opt in (1, 2, 1, 2):
            N.set_option("gemm_s64_rows", rows_opt)
            pair(); torch.cuda.synchronize()
            res.setdefault(rows_opt, []).append((timeit(pair), ys.clone()))
        same = torch.equal(res[1][0][1], res[2][0][1])
        print("C=%4d M=%5d: 64-row tiles (%3d) %6.1f / %6.1f us | 32-row tiles (%3d) %6.1f / %6.1f us per layer pair | bit-identical: %s" %
              (C, M, -(-M // 64) * (C // 80), res[1][0][0], res[1][1][0], -(-M // 32) * (C // 80), res[2][0][0], res[2][1][0], same), flush=True)
