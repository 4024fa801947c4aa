"""Phase timeline of the 18-channel grouped-conv kernel gconv18_shift_kernel (ablation build -DGC_TIMELINE: lane 0 of every wave of the
first 4096 workgroups stamps wall_clock64 at: 0 start, 1 slab written, 2 after the barrier, 3 phase A (shifted tile) done, 4 phase B
weights in registers, 5 phase B blocks done, 6 after the barriers / held-back rows, 7 end of the store phase).
scripts/build_ablation.sh gc_timeline -DGC_TIMELINE && TAL_ASRD_LIB=build/abl/gc_timeline.so python scripts/gconv_timeline.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tal_asrd_amd import ops, _native as N
lib = N.lib()
dev = torch.device("cuda:0")
G = 80
for B, T, cg in ((1, 179991, 10), (1, 89986, 14), (1, 44983, 18)):
    C = G * cg
    x = torch.randn(B, T, C, device=dev)
    w = torch.randn(C, cg, 21, device=dev) / (21 * cg) ** 0.5
    b = torch.randn(C, device=dev)
    wf = ops.pack_gconv_f16x3_weight(w, G)
    xs = ops.split_f16x3(x.view(B * T, C))
    for _ in range(3): ys = ops.gconv_res_split(xs, (B, T, C), wf, b, 0.25, G)
    torch.cuda.synchronize()
    buf = np.zeros(8 * 4 * 4096, dtype=np.uint64)
    fn = ctypes.CDLL(N.LIB_PATH).tal_debug_gconv_timeline
    assert fn(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    t = buf.reshape(8, 4, 4096).astype(np.int64)
    t0 = t[0].min()
    d = (t - t0) * 0.01          # us (100 MHz counter)
    print("cg=%d B=%d T=%d: first 4096 workgroups; kernel start -> last stamp %.1f us" % (cg, B, T, d.max()))
    if cg == 18:
        ph = ["fill (loads + ds_write + zeroing)", "barrier", "phase A (shifted tile)", "phase B weights", "phase B blocks", "barriers + held-back rows", "store phase"]
    else:          # gconv_mfma_kernel: no stamp 3 (2 -> 4 = weight / bias loads)
        t[3] = t[2]
        d[3] = d[2]
        ph = ["fill (loads + ds_write + zeroing)", "barrier", "--", "weight / bias loads", "block loop", "barrier", "store phase"]
    for h in ((0, 1) if cg == 18 else (0,)):
        print("  waves with h = %d (group 0 and 1)" % h if cg == 18 else "  all waves")
        for i in range(7):
            seg = (d[i + 1] - d[i])[h::2].reshape(-1) if cg == 18 else (d[i + 1] - d[i]).reshape(-1)
            print("   %-36s mean %6.2f us   median %6.2f   p90 %6.2f" % (ph[i], seg.mean(), np.median(seg), np.percentile(seg, 90)))
    tot = d[7].max(axis=0) - d[0].min(axis=0)
    print("   workgroup lifetime                   mean %6.2f us   median %6.2f   p90 %6.2f;  starts spread over %.1f us" % (tot.mean(), np.median(tot), np.percentile(tot, 90), d[0].max()))
    # how many of these workgroups are alive at a time (they are the first 4096 of 7040, 512 slots on the chip)
    starts, ends = d[0].min(axis=0), d[7].max(axis=0)
    for q in (0.25, 0.5, 0.75):
        tt = np.quantile(ends, q)
        print("   alive at t = %.1f us: %d" % (tt, int(((starts <= tt) & (ends > tt)).sum())))
