"""One call over a batch of segments against one call per segment (5-minute clips: BASELINE.json configs[3] style).
B = 8 x 300 s: 16.6 M frames/s as one [8, L] call, 12.1 M as eight B = 1 calls (short inputs leave the chip under-filled)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import __graft_entry__ as g
g.build()
from tal_asrd_amd import SDModel, synth
dev = torch.device("cuda:0")
model = SDModel()
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
sd = synth.fill_state_dict(shapes)
own = model.state_dict()
for k, v in sd.items():
    own[k] = torch.from_numpy(v.copy())
model.load_state_dict(own); model.to(dev)
L = 300 * 16000
for B in (1, 4, 8):
    clip = torch.from_numpy(synth.synth_audio_batch(B, L, 1234)).to(dev)
    with torch.no_grad():
        for _ in range(2): feat, ids = model.speaker_ids(clip)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): feat, ids = model.speaker_ids(clip)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("B=%d x 300 s as ONE call: %.2f ms -> %.2f M frames/s  feat %s ids %s" % (B, dt * 1e3, B * (1 + L // 160) / dt / 1e6, tuple(feat.shape), tuple(ids.shape)))
    if B > 1:
        with torch.no_grad():
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                for b in range(B): model.speaker_ids(clip[b:b+1])
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print("   the same as %d separate B=1 calls: %.2f ms -> %.2f M frames/s" % (B, dt * 1e3, B * (1 + L // 160) / dt / 1e6))
