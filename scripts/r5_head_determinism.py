"""Is the diarization head's arg-max the same run after run?  The batch-of-64 call (239k rows: head_argmax_kernel) and the one-item
call with logits (3733 rows), 200 calls each on the same input; every call compared with the first.
python scripts/r5_head_determinism.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import SDModel, synth, ops
dev = torch.device("cuda:0")
m = SDModel()
sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
own = m.state_dict()
for k, v in sd.items():
    own[k] = torch.from_numpy(v.copy())
m.load_state_dict(own); m.to(dev)
gen = torch.Generator(device="cuda").manual_seed(11)
mel = torch.randn(64, 30001, 80, generator=gen, device=dev)
N = int(os.environ.get("REPS", "200"))
with torch.no_grad():
    enc = m.encoder.forward_time_major(mel)
    heads = (m.spk_embed_proj.weight, m.spk_embed_proj.bias, m.spk_logit_proj.weight, m.spk_logit_proj.bias)
    for name, x, want_logits in (("batch of 64, ids only", enc, False), ("item 17 alone, logits + ids", enc[17:18].contiguous(), True),
                                 ("item 17 alone, ids only", enc[17:18].contiguous(), False)):
        f0, l0, i0 = ops.sd_head(x, *heads, want_logits, True)
        torch.cuda.synchronize()
        bad_runs, bad_rows = 0, set()
        fbad = 0
        for k in range(N):
            f, l, i = ops.sd_head(x, *heads, want_logits, True)
            ne = (i != i0)
            if bool(ne.any()):
                bad_runs += 1
                idx = ne.nonzero()
                for r in idx[:8].tolist():
                    bad_rows.add(tuple(r))
                if bad_runs <= 3:
                    r = idx[0].tolist()
                    print("   run %d: %d rows differ, first %s: %d vs %d" % (k, int(ne.sum()), r, int(i[tuple(r)]), int(i0[tuple(r)])), flush=True)
            if not torch.equal(f, f0):
                fbad += 1
        print("%s: %d of %d calls differ from the first in ids (rows seen: %s); features differ in %d calls" % (name, bad_runs, N, sorted(bad_rows)[:10], fbad), flush=True)
    # the encoder itself, run to run
    e0 = m.encoder.forward_time_major(mel[17:18].contiguous())
    nb = 0
    for k in range(50):
        e = m.encoder.forward_time_major(mel[17:18].contiguous())
        nb += int(not torch.equal(e, e0))
    print("encoder on item 17: %d of 50 calls differ from the first" % nb)
