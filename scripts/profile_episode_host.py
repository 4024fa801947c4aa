"""Host-side profile (cProfile) of the sliding-window greedy decode of a 10-minute episode: where the Python time of a decode
step goes beside the C call.  python scripts/profile_episode_host.py [seconds]"""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import ASRModel, synth
from tal_asrd_amd.system import System
from tal_asrd_amd.tokenizer import SynthTokenizer
dev = torch.device("cuda:0")
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
asr = ASRModel("2x", num_speakers=6008, vocab_size=10000, use_speaker_head=True)
sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in asr.state_dict().items()})
own = asr.state_dict()
for k, v in sd.items():
    own[k] = torch.from_numpy(v.copy())
asr.load_state_dict(own); asr.to(dev)
L = int(seconds * 16000)
audio = torch.from_numpy(synth.synth_audio_batch(1, L, 2468).astype(np.float16).astype(np.float32)).to(dev)
system = System(asr, tokenizer=SynthTokenizer(10000))
lens = torch.tensor([L])
system.transcribe_unaligned(audio, lens)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
utts, gen, align = system.transcribe_unaligned(audio, lens)
torch.cuda.synchronize()
pr.disable()
print("steps:", int(gen.shape[1]) - 1)
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(18)
print(out.getvalue())
