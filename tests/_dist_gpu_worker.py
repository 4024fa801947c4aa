"""Worker for tests/test_gpu_distributed.py: 2 ranks (gloo rendezvous, both on the one visible GPU) run ONE reference
call over a [4, L] batch sharded by segment exactly as bench.py's configs[3] workload does -- log-mel without the mean,
the call's (sum, count) all-reduced, subtract, encoder + head, results gathered to rank 0 -- and rank 0 compares the
gathered ids / features with the same call executed in one process."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402
g.build()
from tal_asrd_amd import SDModel, ops, synth  # noqa: E402
from tal_asrd_amd import distributed as D  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = SDModel()
    sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    own = model.state_dict()
    for k, v in sd.items():
        own[k] = torch.from_numpy(v.copy()) if rank == 0 else torch.zeros_like(own[k])   # rank 1 starts with garbage
    model.load_state_dict(own)
    model.to(dev)
    for t in list(model.parameters()) + list(model.buffers()):      # broadcast through host memory (gloo)
        h = t.data.cpu()
        dist.broadcast(h, src=0)
        t.data.copy_(h)
    n_seg, L = 4, 160000 * 3
    lens = [L] * n_seg
    audio = np.concatenate([synth.synth_audio_batch(1, L, 500 + i) for i in range(n_seg)])
    mine = D.shard_indices(n_seg, rank, world, weights=lens)
    batch = torch.from_numpy(audio[mine]).to(dev)
    with torch.no_grad():
        mel, _, st = ops.logmel(model.logmelspec.plan(), batch, eps=model.logmelspec.eps, subtract_mean=False, return_stats=True)
        st = st.cpu()
        mean = D.allreduce_logmel_stats(st).to(dev)
        feat, ids = model.speaker_ids_from_logmel(mel, mean)      # (bench.py --workload segments' own sequence)
        feat_l = {i: feat[k].cpu() for k, i in enumerate(mine)}
        ids_l = {i: ids[k].cpu() for k, i in enumerate(mine)}
        got_feat = D.gather_segments(feat_l, n_seg, dst=0)
        got_ids = D.gather_segments(ids_l, n_seg, dst=0)
        if rank == 0:
            full_feat, full_ids = model.speaker_ids(torch.from_numpy(audio).to(dev))     # the reference call in one process
            for i in range(n_seg):
                assert torch.equal(got_ids[i], full_ids[i].cpu()), i
                err = float((got_feat[i] - full_feat[i].cpu()).abs().max())
                assert err < 2e-4, (i, err)
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()
