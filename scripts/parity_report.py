"""Max deviation of the SD path from the vectors recorded from the reference (tests/golden/sd_5min.npz) -- run once as
is (fp16x3 dense layers) and once with TAL_OPTIONS=tds_exact_f32 (pure fp32 dense layers)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import SDModel, synth, ops
dev = torch.device("cuda:0")
gold = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "sd_5min.npz"))
m = SDModel()
sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
own = m.state_dict()
for k, v in sd.items():
    own[k] = torch.from_numpy(v.copy())
m.load_state_dict(own); m.to(dev)
audio = torch.from_numpy(synth.synth_audio_batch(int(gold["batch"]), int(gold["audio_len"]), int(gold["audio_seed"]))).to(dev)
with torch.no_grad():
    enc = m.encode(audio, None)
    logits = m.decode(enc)
eo = enc["encoder_out"]
print("mode: %s" % ("fp32 dense layers" if "tds_exact_f32" in os.environ.get("TAL_OPTIONS", "") else "fp16x3 dense layers"))
print("  encoder_out  max |err| vs reference sample: %.3e" % np.abs(eo[:, gold["enc_rows"]].cpu().numpy() - gold["enc_sample"]).max())
print("  logits       max |err| vs reference sample: %.3e" % np.abs(logits[:, gold["logit_rows"]].cpu().numpy() - gold["logit_sample"]).max())
ids = ops.argmax_rows(logits).cpu().numpy()
print("  speaker ids differing from the reference: %d of %d" % (int((ids != gold["ids"]).sum()), ids.size))
