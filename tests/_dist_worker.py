"""Worker for tests/test_distributed_cpu.py: run under torch.distributed.run with
world_size 2 on CPU (gloo).  Exercises the N>1 sharding / broadcast / gather / mean
all-reduce logic of tal_asrd_amd.distributed with a stand-in for the per-segment GPU work."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tal_asrd_amd import distributed as D  # noqa: E402


def fake_segment_result(i, length):
    """Deterministic stand-in for (ids, feat) of segment i with `length` encoder frames."""
    g = torch.Generator().manual_seed(100 + i)
    return torch.randint(0, 6008, (length,), generator=g, dtype=torch.int32), torch.randn(length, 8, generator=g)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == 2

    # 1) weight broadcast: rank 1 starts with garbage, ends with rank 0's values
    lin = torch.nn.Linear(4, 3)
    with torch.no_grad():
        lin.weight.fill_(float(rank + 1))
        lin.bias.fill_(float(rank + 1))
    D.broadcast_module(lin, src=0)
    assert float(lin.weight.min()) == 1.0 and float(lin.bias.max()) == 1.0

    # 2) sharding: disjoint cover, deterministic, length-balanced
    lengths = [3733, 358, 3733, 1200, 44983, 90, 358, 2000]
    n = len(lengths)
    mine = D.shard_indices(n, rank, world, weights=lengths)
    other = D.shard_indices(n, 1 - rank, world, weights=lengths)
    assert sorted(mine + other) == list(range(n)) and not set(mine) & set(other)
    assert D.shard_indices(5, rank, world) == list(range(rank, 5, world))
    loads = [sum(lengths[i] for i in D.shard_indices(n, r, world, weights=lengths)) for r in range(world)]
    assert max(loads) <= max(lengths) + min(loads)

    # 3) variable-length gather of per-segment outputs to rank 0, item order restored
    ids_local = {i: fake_segment_result(i, lengths[i])[0] for i in mine}
    feat_local = {i: fake_segment_result(i, lengths[i])[1] for i in mine}
    ids = D.gather_segments(ids_local, n, dst=0)
    feat = D.gather_segments(feat_local, n, dst=0)
    if rank == 0:
        for i in range(n):
            want_ids, want_feat = fake_segment_result(i, lengths[i])
            assert torch.equal(ids[i], want_ids), i
            assert torch.equal(feat[i], want_feat), i
    else:
        assert ids is None and feat is None

    # 3b) the planned form: plan built once, features + ids in ONE collective per call (ids bit-cast into an extra column),
    #     re-used for a second call with other values
    plan = D.SegmentGather({i: lengths[i] for i in mine}, n, trailing=(9,), dtype=torch.float32, device="cpu", dst=0)
    for rep in range(2):
        packed_local = {i: D.pack_feat_ids(feat_local[i] + rep, ids_local[i]) for i in mine}
        got = plan.gather(packed_local)
        if rank == 0:
            for i in range(n):
                want_ids, want_feat = fake_segment_result(i, lengths[i])
                f, d = D.unpack_feat_ids(got[i])
                assert torch.equal(f, want_feat + rep) and torch.equal(d, want_ids), (rep, i)
        else:
            assert got is None
    # 3c) a rank WITHOUT items needs no `like`, and learns shape / dtype from the others
    solo = D.gather_segments({0: torch.full((3, 2), 7.0)} if rank == 1 else {}, 1, dst=0)
    if rank == 0:
        assert solo[0].shape == (3, 2) and float(solo[0].min()) == 7.0
    # 3d) inconsistent calls fail on EVERY rank (nobody is left inside a collective): an item nobody holds, an item two
    #     ranks claim
    for bad in ({0: 4} if rank == 0 else {}, {0: 4, 1: 4} if rank == 0 else {1: 4}):
        try:
            D.SegmentGather(bad, 2 if rank == 0 or len(bad) else 2, trailing=(1,) if bad else None,
                            dtype=torch.float32 if bad else None, device="cpu")
        except RuntimeError as e:
            assert "SegmentGather" in str(e)
        else:
            raise AssertionError("inconsistent gather plan accepted on rank %d" % rank)
    dist.barrier()

    # 4) one reference "call" split across ranks: the global log-mel mean via (sum, count)
    full = torch.arange(24, dtype=torch.float64).reshape(2, 3, 4)   # the [B, T, 80]-like tensor of one call
    part = full[rank]
    stats = torch.tensor([float(part.sum()), float(part.numel())], dtype=torch.float64)
    mean = D.allreduce_logmel_stats(stats)
    assert abs(float(mean) - float(full.mean())) < 1e-6

    # 5) one clip over both ranks: tiles with their halo through a stand-in encoder (the oracle's TDS on CPU), stitched on
    #    rank 0, equal to the whole-clip result
    from oracle import tal_oracle as O
    from tests.test_oracle_golden import _fill, _tds_shapes

    class OracleEncoder:
        sizes, depths = [8, 16, 24, 32], [1, 1, 2]

        def __init__(self):
            self.sd = _fill(_tds_shapes(self.sizes, self.depths, 8), "tds_small.")

        def forward_time_major(self, x):
            return O.tds_forward(x.permute(0, 2, 1), self.sd, prefix="", depths=tuple(self.depths), groups=8).permute(0, 2, 1).contiguous()

    enc = OracleEncoder()
    mel = torch.randn(1, 2100, 8, generator=torch.Generator().manual_seed(9))
    got = D.encode_clip_sharded(enc, mel, out_tile=48, dst=0)
    if rank == 0:
        want = enc.forward_time_major(mel)
        assert got.shape == want.shape and float((got - want).abs().max()) < 5e-6
    else:
        assert got is None

    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()
