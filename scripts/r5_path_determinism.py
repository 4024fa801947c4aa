"""The whole SD path call after call on one input -- features and ids of SDModel.speaker_ids compared bit for bit with the first call's:
the 1-hour clip (configs[2]), a batch of 64 five-minute segments (configs[3]), a 5-minute and a 30-second clip.
python scripts/r5_path_determinism.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import SDModel, synth
dev = torch.device("cuda:0")
m = SDModel()
sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
own = m.state_dict()
for k, v in sd.items():
    own[k] = torch.from_numpy(v.copy())
m.load_state_dict(own); m.to(dev)
for name, B, sec, n in (("1-hour clip", 1, 3600, 100), ("64 x 5 min", 64, 300, 40), ("5-minute clip", 1, 300, 300), ("30-second clip", 1, 30, 500)):
    x = torch.from_numpy(synth.synth_audio_batch(B, sec * 16000, 4321)).to(dev)
    f0, i0 = m.speaker_ids(x)
    bad_f = bad_i = 0
    for _ in range(n):
        f, i = m.speaker_ids(x)
        bad_f += int(not torch.equal(f, f0))
        bad_i += int(not torch.equal(i, i0))
    print("%s: %d calls: features differ from the first call's in %d, ids in %d" % (name, n, bad_f, bad_i), flush=True)
    del x, f0, i0
    torch.cuda.empty_cache()
