"""ASRModel.encode (encoder_out, speaker_out) and ASRModel.decode / decode_spk call after call on one input, bit for bit.
python scripts/r5_asr_determinism.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import ASRModel, synth
dev = torch.device("cuda:0")
m = ASRModel("2x", num_speakers=6008, vocab_size=10000, use_speaker_head=True)
sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
own = m.state_dict()
for k, v in sd.items():
    own[k] = torch.from_numpy(v.copy())
m.load_state_dict(own); m.to(dev).eval()
with torch.no_grad():
    for name, B, sec, n in (("1-hour episode", 1, 3600, 60), ("8 x 5 min", 8, 300, 60), ("30-second clip", 1, 30, 300)):
        x = torch.from_numpy(synth.synth_audio_batch(B, sec * 16000, 99)).to(dev).half()
        lens = torch.full((B,), sec * 16000, dtype=torch.int64)
        e0 = m.encode(x, lens)
        bad = 0
        for _ in range(n):
            e = m.encode(x, lens)
            bad += int(not (torch.equal(e["encoder_out"], e0["encoder_out"]) and torch.equal(e["speaker_out"], e0["speaker_out"])))
        print("encode, %s: %d of %d calls differ from the first" % (name, bad, n), flush=True)
    x = torch.from_numpy(synth.synth_audio_batch(2, 30 * 16000, 5)).to(dev).half()
    enc = m.encode(x, torch.tensor([480000, 400000]))
    for U in (1, 7, 64, 200):
        y = torch.randint(3, 10000, (2, U), device=dev)
        l0, s0 = m.decode(y, enc), m.decode_spk(y, enc)
        bad = 0
        for _ in range(200):
            bad += int(not (torch.equal(m.decode(y, enc), l0) and torch.equal(m.decode_spk(y, enc), s0)))
        print("decode + decode_spk, prefix %d: %d of 200 calls differ from the first" % (U, bad), flush=True)
