"""Phase timeline of the one-launch decode step (ablation build -DPS_TIMELINE): workgroups 0 and G - 1 stamp wall_clock64 at every
phase barrier: 0 body done, 1 stores drained + workgroup barrier, 2 released (arrival counter complete), 3 acquire fence done.
scripts/build_ablation.sh ps_timeline -DPS_TIMELINE && TAL_ASRD_LIB=build/abl/ps_timeline.so python scripts/decode_persist_timeline.py [U] [G]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tal_asrd_amd import ASRModel, synth, _native as N
from tal_asrd_amd.system import _GreedySession
dev = torch.device("cuda:0")
m = ASRModel("2x", num_speakers=6008, vocab_size=10000, use_speaker_head=True)
sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
own = m.state_dict()
for k, v in sd.items():
    own[k] = torch.from_numpy(v.copy())
m.load_state_dict(own); m.to(dev)
win = {"encoder_out": torch.randn(1, 357, 512, device=dev), "encoder_padding_mask": torch.zeros(1, 357, dtype=torch.bool, device=dev)}
gen_dev = torch.randint(3, 10000, (1024,), device=dev)
sess = _GreedySession(m, gen_dev, 512)
sess.set_window(win)
U = int(sys.argv[1]) if len(sys.argv) > 1 else 20
G = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N.set_option("decode_persist", 1)
N.set_option("decode_persist_wgs", G)
for _ in range(20): sess.step(0, U)
torch.cuda.synchronize()
buf = np.zeros(2 * 64 * 4, dtype=np.uint64)
fn = ctypes.CDLL(N.LIB_PATH).tal_debug_ps_timeline
assert fn(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(2, 64, 4).astype(np.int64)
names = ["embed"] + [n for l in range(4) for n in ("qkv", "self-attn", "sa-out", "ca-q", "cross-attn", "ca-out", "ffn1", "ffn2")]
print("prefix %d tokens, %d workgroups; us: body (from the previous fence) | drain + workgroup barrier | wait for the others | acquire fence" % (U, G))
for who in (0, 1):
    tt = t[who]
    print(" workgroup %s" % ("0" if who == 0 else "G - 1"))
    tot = np.zeros(4)
    for p in range(len(names)):
        prev = tt[p - 1][3] if p > 0 else tt[0][0]
        seg = np.array([tt[p][0] - prev, tt[p][1] - tt[p][0], tt[p][2] - tt[p][1], tt[p][3] - tt[p][2]]) * 0.01
        tot += seg
        print("   %-10s %6.2f %6.2f %6.2f %6.2f" % (names[p], *seg))
    print("   %-10s %6.2f %6.2f %6.2f %6.2f   (sum %.1f us, first stamp -> last fence %.1f us)" % ("total", *tot, tot.sum(), (tt[len(names) - 1][3] - tt[0][0]) * 0.01))
