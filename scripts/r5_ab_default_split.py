import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import ASRModel, synth
from tal_asrd_amd.system import System
from tal_asrd_amd.tokenizer import SynthTokenizer
dev = torch.device("cuda:0")
m = ASRModel("2x", num_speakers=6008, vocab_size=10000, use_speaker_head=True)
sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
own = m.state_dict()
for k, v in sd.items(): own[k] = torch.from_numpy(v.copy())
m.load_state_dict(own); m.to(dev)
system = System(m, tokenizer=SynthTokenizer(10000))
L = 3600 * 16000
eps = [(torch.from_numpy(synth.synth_audio_batch(1, L, 2469 + i).astype(np.float16).astype(np.float32)).pin_memory(), torch.tensor([L])) for i in range(8)]
for rep in range(3):
    for k, gsz in ((4, 2), (None, None)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st = {}
        many = system.transcribe_unaligned_many(eps, streams=k, group=gsz, stats=st)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(k, gsz, "%.3f s" % dt, st, flush=True)
