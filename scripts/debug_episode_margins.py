"""Where does the GPU decode of the 1-hour episode differ from the reference fixture, and how close was the call?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import ASRModel, synth
from tal_asrd_amd import system as S
from tal_asrd_amd.system import System
from tal_asrd_amd.tokenizer import SynthTokenizer

dev = torch.device("cuda:0")
fx = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "episode_1h.npz"))
m = ASRModel("2x", num_speakers=6008, vocab_size=10000, use_speaker_head=True)
sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
own = m.state_dict()
for k, v in sd.items():
    own[k] = torch.from_numpy(v.copy())
m.load_state_dict(own); m.to(dev)
L = int(fx["audio_len"])
audio = torch.from_numpy(synth.synth_audio_batch(1, L, int(fx["audio_seed"])).astype(np.float16).astype(np.float32)).to(dev)
margins = []
orig = S.asr_decode
def spy(*a, **k):
    lg = orig(*a, **k)
    t = torch.topk(lg[0], 3)
    margins.append((t.values.cpu().numpy(), t.indices.cpu().numpy()))
    return lg
S.asr_decode = spy
system = System(m, tokenizer=SynthTokenizer(10000))
gen, al = system.generate_unaligned(audio, torch.ones(1, 1, dtype=torch.long, device=dev), torch.tensor([L]))
gen = gen.cpu().numpy()[0]
want = fx["generated"][0]
print("len", len(gen), len(want), "steps recorded", len(margins))
bad = np.nonzero(gen[:len(want)] != want[:len(gen)])[0]
print("mismatch positions", bad)
# margins are per decode step, not per surviving token; list the closest calls of the whole run
allm = np.array([v[0] - v[1] for v, _ in margins])
order = np.argsort(allm)[:12]
for o in order:
    print("step %d margin %.3e top3 %s vals %s" % (o, allm[o], margins[o][1], margins[o][0]))
for b in bad:
    print("pos", b, "got", gen[b], "want", want[b], "context got", gen[b-3:b+3], "want", want[b-3:b+3])
print("hist of margins: <1e-5 %d, <1e-4 %d, <1e-3 %d of %d" % ((allm < 1e-5).sum(), (allm < 1e-4).sum(), (allm < 1e-3).sum(), len(allm)))
