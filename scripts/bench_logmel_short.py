import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import SDModel, synth
dev = torch.device("cuda:0")
m = SDModel().to(dev)
for sec in (10, 30, 60, 300):
    x = torch.from_numpy(synth.synth_audio_batch(1, sec * 16000, 5)).to(dev)
    for _ in range(5): m.extract_features(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): m.extract_features(x)
    e1.record(); torch.cuda.synchronize()
    print("%d s: extract_features (log-mel + mean + subtract) %.1f us per call" % (sec, e0.elapsed_time(e1) * 10))
