"""A / B on one box, interleaved: SDModel.speaker_ids on a 1-hour clip (and a 5-minute one) with the encoder output handed to the
head in the hi / lo split form (the default) and as fp32 (the round-4 head: fp32 1440 -> 128 embedding layer).
python scripts/r5_ab_split_head.py"""
import os, sys, time
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
split = type(m)._embed_split


def timed(x, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        out = m.speaker_ids(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


for sec, n in ((3600.0, 10), (300.0, 40)):
    x = torch.from_numpy(synth.synth_audio_batch(1, int(sec * 16000), 1234)).to(dev)
    res = {"split": [], "fp32": []}
    outs = {}
    with torch.no_grad():
        for rnd in range(5):
            for name in ("split", "fp32"):
                m._embed_split = (lambda: None) if name == "fp32" else (lambda: split(m))
                if rnd == 0:
                    timed(x, 3)
                ms, outs[name] = timed(x, n)
                res[name].append(ms)
    same = bool(torch.equal(outs["split"][1], outs["fp32"][1]))
    dfeat = float((outs["split"][0] - outs["fp32"][0]).abs().max())
    for name in ("split", "fp32"):
        print("%.0f s clip, head input %-5s: %s ms per call (5 interleaved rounds of %d calls), best %.3f" %
              (sec, name, " ".join("%.3f" % v for v in res[name]), n, min(res[name])), flush=True)
    print("   ids identical: %s, max |feature difference| %.2e" % (same, dfeat), flush=True)
