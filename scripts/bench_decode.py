"""Decode-loop timing: System.generate_unaligned on a synthetic clip (steps/s), and the cost of
one decode step as a function of the prefix length."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import ASRModel, synth
from tal_asrd_amd.system import System
from tal_asrd_amd.decoder import asr_decode

dev = torch.device("cuda:0")
m = ASRModel("2x", num_speakers=6008, vocab_size=10000, use_speaker_head=True)
sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
own = m.state_dict()
for k, v in sd.items():
    own[k] = torch.from_numpy(v.copy())
m.load_state_dict(own); m.to(dev)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2400000
audio = torch.from_numpy(synth.synth_audio_batch(1, L, 4321)).to(dev)
s = System(m)
for it in (50, 260):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gen, al = s.generate_unaligned(audio, torch.ones(1, 1, dtype=torch.long, device=dev), torch.tensor([L]), max_iters=it)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("generate_unaligned: %d steps in %.3f s -> %.1f steps/s (%.2f ms/step), %d tokens" % (it, dt, it / dt, 1e3 * dt / it, gen.shape[1]))
enc = m.encode(audio, torch.tensor([L]))
win = {"encoder_out": enc["encoder_out"][:, :357].contiguous(), "encoder_padding_mask": enc["encoder_padding_mask"][:, :357].contiguous()}
for U in (1, 16, 64, 128, 256, 512):
    y = torch.randint(0, 10000, (1, U), device=dev)
    for _ in range(3): asr_decode(m, y, win, causal=False, last_only=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n): asr_decode(m, y, win, causal=False, last_only=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("decode step U=%3d S=357: %.3f ms" % (U, dt * 1e3))
