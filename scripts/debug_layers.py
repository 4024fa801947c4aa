"""Debug aid (not a test): per-layer comparison of the HIP TDS ops against torch ops on the GPU."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from tal_asrd_amd import SDModel, synth, ops

T = int(sys.argv[1]) if len(sys.argv) > 1 else 30001
dev = torch.device("cuda:0")
m = SDModel()
sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
own = m.state_dict()
for k, v in sd.items():
    own[k] = torch.from_numpy(v.copy())
m.load_state_dict(own); m.to(dev)
g = torch.Generator().manual_seed(5)
x = torch.randn(1, T, 80, generator=g).to(dev)
with torch.no_grad():
    full = m.encoder.forward_time_major(x)
    ref = x.permute(0, 2, 1)
    cur = x
    for i, stage in enumerate(m.encoder.blocks):
        down, chain = stage[0], stage[1]
        ref = F.conv1d(ref, down.weight, down.bias, stride=2, groups=80)
        cur = ops.gconv_s2(cur, down.packed(), down.bias, down.out_channels, 80)
        print("stage", i, "down", tuple(cur.shape), float((cur.permute(0, 2, 1) - ref).abs().max()))
        for j, blk in enumerate(chain):
            gc = blk.conv[0]
            rw = float(blk.resweight)
            r1 = ref + rw * F.relu(F.conv1d(ref, gc.weight, gc.bias, padding=10, groups=80))
            c1 = ops.gconv_res(cur, gc.packed(), gc.bias, rw, 80)
            e1 = float((c1.permute(0, 2, 1) - r1).abs().max())
            h_ref = F.relu(F.conv1d(r1, blk.fc[0].weight, blk.fc[0].bias))
            h = ops.linear(c1, blk.fc[0].weight, blk.fc[0].bias, mode=1)
            e2 = float((h.permute(0, 2, 1) - h_ref).abs().max())
            ref = r1 + rw * F.conv1d(h_ref, blk.fc[3].weight, blk.fc[3].bias)
            cur = ops.linear(h, blk.fc[3].weight, blk.fc[3].bias, mode=2, res=c1, alpha=rw)
            e3 = float((cur.permute(0, 2, 1) - ref).abs().max())
            print("  block", i, j, "gconv %.2e fc0 %.2e fc3 %.2e" % (e1, e2, e3))
    print("stepwise vs torch:", float((cur.permute(0, 2, 1) - ref).abs().max()))
    print("tal_tds_fwd vs stepwise:", float((full - cur).abs().max()))
