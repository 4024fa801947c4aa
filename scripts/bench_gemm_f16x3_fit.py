"""fp16x3 dense-layer kernel on pre-split operands: per-tile fixed cost vs per-K-step cost (whole rounds only, K swept;
time per round = c0 + c1 * K/32) -- how much of a launch is tile set-up + epilogue at the K of this path (800-1440).
Cases: the TDS block's two layers as tal_tds_fwd runs them (guarded, split-form output / residual) and the generic forms."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import ops, _native as N

dev = torch.device("cuda:0")
lib = N.lib()
Nout = int(os.environ.get("FIT_N", "1440"))
M = 128 * 512
rounds = (M // 128) * (Nout // 160) / 512.0
flag = torch.zeros(16, dtype=torch.int32, device=dev)
cases = (("fc0: relu, split out, guarded", 1, 1, True, 0), ("fc1: split residual, split out, guarded", 2, 1, True, 1),
         ("relu, split out, unguarded", 1, 1, False, 0), ("plain fp32 out", 0, 0, False, 0))
Ks = (320, 640, 800, 1120, 1440, 2880, 5760)
best = {c[0]: {} for c in cases}
for K in Ks:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(Nout, K, device=dev) / K ** 0.5
    b = torch.randn(Nout, device=dev)
    xs, wsp = ops.split_f16x3(x), ops.split_f16x3(w)
    res = ops.split_f16x3(torch.randn(M, Nout, device=dev))
    y = torch.empty(M, Nout, device=dev)
    nws = lib.tal_linear_workspace_bytes(M, Nout, K)
    ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)
    for rep in range(4):           # cases interleaved, best of 4: clocks drift with what ran before
        for name, mode, out_split, guarded, res_split in cases:
            def run():
                N.check(lib.tal_linear_f16x3_guarded_fwd(N.ptr(xs), N.ptr(wsp), N.ptr(b), N.ptr(res) if res_split else None, res_split,
                                                         0.3, mode, M, Nout, K, N.ptr(y), out_split, N.ptr(flag) if guarded else None,
                                                         N.ptr(ws), nws, N.stream_handle()), "tal_linear_f16x3_guarded_fwd")
            for _ in range(2): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10
            e0.record()
            for _ in range(n): run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            best[name][K] = min(best[name].get(K, 1e9), ms)
for name, *_ in cases:
    pts = [(K // 32, best[name][K] * 1e3 / rounds) for K in Ks]
    print(name + ": " + "  ".join("K=%d %.1f us/round (%.0f TF)" % (K, best[name][K] * 1e3 / rounds, 2.0 * M * Nout * K / best[name][K] / 1e9) for K in Ks))
    a = np.array(pts)
    c1, c0 = np.polyfit(a[:, 0], a[:, 1], 1)
    print("  fit: per-round fixed %.1f us, per K-step %.3f us" % (c0, c1), flush=True)
