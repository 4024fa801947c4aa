"""Stage timeline of the fast-transform log-mel kernel (ablation build -DLM_TIMELINE): workgroup 100 stamps the wall clock after
every stage of its first 8 blocks of 12 frames.
scripts/build_ablation.sh lm_timeline -DLM_TIMELINE && TAL_ASRD_LIB=build/abl/lm_timeline.so python scripts/r5_logmel_timeline.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tal_asrd_amd import SDModel, synth, ops, _native as N
dev = torch.device("cuda:0")
m = SDModel().to(dev)
lm = m.logmelspec
x = torch.from_numpy(synth.synth_audio_batch(1, 3600 * 16000, 99)).to(dev)
for _ in range(3):
    ops.logmel(lm.plan(), x, eps=lm.eps, subtract_mean=False)
torch.cuda.synchronize()
buf = np.zeros(64, dtype=np.uint64)
fn = ctypes.CDLL(N.LIB_PATH).tal_debug_logmel_timeline
assert fn(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(8, 8).astype(np.int64)
names = ["samples -> LDS + barrier", "prefetch issue + pass 1 + barrier", "pass 2 + barrier + Z store + barrier", "power + barrier", "mel + log + barrier",
         "output store + barrier"]
print("us per stage, blocks 1..7 of workgroup 100 (1-hour clip)")
for it in range(1, 8):
    seg = [(t[it][i + 1] - t[it][i]) * 0.01 for i in range(6)]
    print("  block %d: " % it + "  ".join("%s %.2f" % (n, v) for n, v in zip(names, seg)) + "  | total %.2f (loop top to loop top %.2f)" % (
        (t[it][6] - t[it][0]) * 0.01, (t[it][0] - t[it - 1][0]) * 0.01))
