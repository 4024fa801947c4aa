"""The TDS block's dense pair on short inputs: 64 x 80 tiles (csrc/gemm_s64.hip) against the K-sliced 128 x 160 / 256 x 160 path.
Per shape: max error of both against float64, time per layer pair.  python scripts/bench_gemm_short.py [rows ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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

rows = [int(a) for a in sys.argv[1:]] or [200, 376, 751, 1501, 3000, 3751, 7501, 15001]
for C in (800, 1120, 1440):
    for M in rows:
        gen = torch.Generator().manual_seed(M + C)
        x = torch.randn(M, C, generator=gen).to(dev)
        w0 = (torch.randn(C, C, generator=gen) / C ** 0.5).to(dev)
        b0 = torch.randn(C, generator=gen).to(dev)
        w1 = (torch.randn(C, C, generator=gen) / C ** 0.5).to(dev)
        b1 = torch.randn(C, generator=gen).to(dev)
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
        def decode(buf):
            h = buf.view(torch.float16).reshape(M, C // 32, 64).float()
            return (h[:, :, :32] + h[:, :, 32:] / 2048.0).reshape(M, C)
        xd = decode(xs.view(torch.uint8).reshape(-1)).double()
        h = torch.relu(xd @ w0.double().t() + b0.double())
        out = {}
        for name, lim in (("sliced", 0), ("s64", 1 << 20)):
            N.set_option("gemm_s64_below", lim)
            pair(); torch.cuda.synchronize()
            hd = decode(hs).double()
            ref = xd + 0.3 * (hd @ w1.double().t() + b1.double())      # (second layer against its own input)
            e0 = float((hd - h).abs().max()); e1 = float((decode(ys).double() - ref).abs().max())
            out[name] = (timeit(pair), e0, e1)
        N.set_option("gemm_s64_below", 2)
        print("C=%4d M=%6d: sliced %7.1f us (err %.1e %.1e) | 64x80 tiles %7.1f us (err %.1e %.1e) | flag %d" %
              (C, M, out["sliced"][0], out["sliced"][1], out["sliced"][2], out["s64"][0], out["s64"][1], out["s64"][2], int(flag[0])), flush=True)
