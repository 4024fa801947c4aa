"""Matrix-core grouped convs on split-form activations (the product path), 1-hour shapes and the 8-segment batch: XCD-aware
1-D grid against the plain (time tile, group block, item) grid (option gconv_grid_xyz); for 18 channels also the time-shift-
packed kernel against the two-M-tile kernel (gconv_no_shift18).  Interleaved samples of 20 launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import ops, _native as N
dev = torch.device("cuda:0")
G = 80
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def ab(label, fn, variants):
    res = {k: [] for k in variants}
    for rep in range(3):
        for k, opts in variants.items():
            for o, v in opts.items(): N.set_option(o, v)
            res[k].append(timeit(fn))
            for o in opts: N.set_option(o, 4 if o == "gconv_short_below" else 0)
    print(label + ": " + " | ".join("%s %.4f ms" % (k, min(v)) for k, v in res.items()), flush=True)
for B, scale in ((1, 1.0), (8, 1.0 / 12.05)):
    for T, cg in ((179991, 10), (89986, 14), (44983, 18)):
        T = int(T * scale)
        C = G * cg
        x = torch.randn(B, T, C, device=dev)
        w = torch.randn(C, cg, 21, device=dev) / (21 * cg) ** 0.5
        b = torch.randn(C, device=dev)
        wf = ops.pack_gconv_f16x3_weight(w, G)
        xs = ops.split_f16x3(x.view(B * T, C))
        variants = {"xcd order": {}, "plain grid": {"gconv_grid_xyz": 1}, "256-step tiles": {"gconv_long_tt": 256}, "128-step tiles": {"gconv_long_tt": 128}, "64-step tiles": {"gconv_short_below": 1000000}}
        if cg == 18:
            variants = {"shift + xcd": {}, "shift, plain grid": {"gconv_grid_xyz": 1}, "two tiles + xcd": {"gconv_no_shift18": 1},
                        "two tiles, plain grid (round 3)": {"gconv_no_shift18": 1, "gconv_grid_xyz": 1}, "shift, 64-step tiles": {"gconv_short_below": 1000000}}
        ab("res  cg=%2d B=%d T=%6d" % (cg, B, T), lambda: ops.gconv_res_split(xs, (B, T, C), wf, b, 0.25, G), variants)
    for T, cin, cout in ((179991, 10, 14), (89986, 14, 18)):
        T = int(T * scale)
        x = torch.randn(B, T, G * cin, device=dev)
        w = torch.randn(G * cout, cin, 21, device=dev) / (21 * cin) ** 0.5
        b = torch.randn(G * cout, device=dev)
        wf = ops.pack_gconv_f16x3_weight(w, G, stride=2)
        xs = ops.split_f16x3(x.view(B * T, G * cin))
        ab("s2 %d->%d B=%d T=%6d" % (cin, cout, B, T), lambda: ops.gconv_s2_split(xs, (B, T, G * cin), True, wf, b, G * cout, G),
           {"xcd order": {}, "plain grid": {"gconv_grid_xyz": 1}, "64-step tiles": {"gconv_short_below": 1000000}})
