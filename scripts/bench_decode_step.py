"""One decode step (asr_decode, last position only) repeated, for a kernel-sequence trace."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import ASRModel, synth
from tal_asrd_amd.decoder import asr_decode
dev = torch.device("cuda:0")
m = ASRModel("2x", num_speakers=6008, vocab_size=10000, use_speaker_head=True)
sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
own = m.state_dict()
for k, v in sd.items():
    own[k] = torch.from_numpy(v.copy())
m.load_state_dict(own); m.to(dev)
U = int(sys.argv[1]) if len(sys.argv) > 1 else 64
win = {"encoder_out": torch.randn(1, 357, 512, device=dev), "encoder_padding_mask": torch.zeros(1, 357, dtype=torch.bool, device=dev)}
y = torch.randint(0, 10000, (1, U), device=dev)
for _ in range(5): asr_decode(m, y, win, causal=False, last_only=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): asr_decode(m, y, win, causal=False, last_only=True)
torch.cuda.synchronize(); print("U=%d: %.3f ms/step" % (U, (time.perf_counter() - t0) / 20 * 1e3))
