"""The diarization head's arg-max launch (head_argmax_kernel + the partial merge) on 44,983 and 238,912 rows: time per call and run-to-run
identity of the ids.  Run once with the product library and once with TAL_ASRD_LIB=build/abl/head_nowait.so
(scripts/build_ablation.sh head_nowait -DHEAD_NO_DMA_WAIT: the kernel as it was before round 5's fix)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
if not os.environ.get("TAL_ASRD_LIB"):
    g.build()
from tal_asrd_amd import SDModel, synth, ops
dev = torch.device("cuda:0")
m = SDModel()
sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
own = m.state_dict()
for k, v in sd.items():
    own[k] = torch.from_numpy(v.copy())
m.load_state_dict(own); m.to(dev)
heads = (m.spk_embed_proj.weight, m.spk_embed_proj.bias, m.spk_logit_proj.weight, m.spk_logit_proj.bias)
gen = torch.Generator(device="cuda").manual_seed(5)
for rows in (44983, 238912):
    x = torch.randn(1, rows, 1440, generator=gen, device=dev)
    with torch.no_grad():
        f0, _, i0 = ops.sd_head(x, *heads, False, True)
        torch.cuda.synchronize()
        bad = 0
        t0 = time.perf_counter()
        n = 200
        for _ in range(n):
            f, _, i = ops.sd_head(x, *heads, False, True)
            bad += int(not torch.equal(i, i0))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    print("%s: %6d rows: %.1f us per head call (embedding layer + arg-max + merge + the comparison), %d of %d calls with ids other than the first call's"
          % (os.environ.get("TAL_ASRD_LIB", "product library"), rows, dt * 1e6, bad, n), flush=True)
