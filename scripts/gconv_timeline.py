"""Phase timeline of the grouped-conv kernel (ablation build -DGC_ABL_TIMELINE: thread 0 of each of the first 4096 workgroups stamps
wall_clock64 at: 0 start, 1 slab rows written, 2 pads zeroed, 3 after the barrier, 4 weights in registers, 5 end of the block loop, 6 after the barrier, 7 end of the store phase).
TAL_ASRD_LIB=build/abl/gc_timeline.so python scripts/gconv_timeline.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tal_asrd_amd import ops, _native as N
lib = N.lib()
dev = torch.device("cuda:0")
G = 80
for T, cg in ((179991, 10), (89986, 14), (44983, 18)):
    C = G * cg
    x = torch.randn(1, T, C, device=dev)
    w = torch.randn(C, cg, 21, device=dev) / (21 * cg) ** 0.5
    b = torch.randn(C, device=dev)
    wf = ops.pack_gconv_f16x3_weight(w, G)
    xs = ops.split_f16x3(x.view(T, C))
    for _ in range(3): ys = ops.gconv_res_split(xs, (1, T, C), wf, b, 0.25, G)
    torch.cuda.synchronize()
    buf = np.zeros(8 * 4096, dtype=np.uint64)
    assert lib.tal_debug_gconv_timeline(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    t = buf.reshape(8, 4096).astype(np.int64)
    t0 = t[0].min()
    d = (t - t0) * 0.01          # us (100 MHz counter)
    order = np.argsort(d[0])
    print("cg=%d T=%d: first 4096 workgroups; kernel start -> last stamp %.1f us" % (cg, T, d.max()))
    ph = ["fill (loads + ds_write)", "pad zeroing", "barrier", "weight / bias loads", "block loop", "barrier", "store phase"]
    for i in range(7):
        if t[i + 1].max() == 0: continue
        seg = d[i + 1] - d[i]
        print("   %-26s mean %6.2f us   median %6.2f   p90 %6.2f" % (ph[i], seg.mean(), np.median(seg), np.percentile(seg, 90)))
    last = 7 if t[7].max() else 5
    tot = d[last] - d[0]
    print("   workgroup lifetime          mean %6.2f us   median %6.2f   p90 %6.2f;  starts spread over %.1f us" % (tot.mean(), np.median(tot), np.percentile(tot, 90), d[0].max()))
