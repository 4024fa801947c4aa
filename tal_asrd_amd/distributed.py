"""Multi-GPU inference by segment sharding.

The path shards naturally: audio segments / episodes are independent units (the reference
treats every batch independently, tal/asr/transcribe.py:124-162, tal/asr/system.py:625-742,
and `generate_unaligned` is one episode at a time).  One process per GPU, whole segments
only, no collective on the data path.  The only exchanges are

  * a one-time broadcast of the weights from rank 0 (RCCL over xGMI: ~274 MB for the full
    ASR model, a few ms per hop -- irrelevant to steady state);
  * the gather of per-segment results (speaker ids + 128-d features, or logits when parity
    is being checked) to rank 0;
  * optionally ONE scalar all-reduce when a single reference "call" (one LogMelSpec.forward
    over a [B, L] batch, whose subtracted mean couples all B items, tal/asr/models.py:52) is
    split across ranks: the (sum, count) pair of the log-mel tensor.

Everything here works on any torch.distributed backend: "nccl" (= RCCL) on the GPU box,
"gloo" in the CPU tests (tests/test_distributed_cpu.py, world_size 2).
"""
import torch
import torch.distributed as dist

from . import tiling


def shard_indices(n_items: int, rank: int, world: int, weights=None):
    """Indices of the items `rank` processes.

    Without weights: round-robin (item i -> rank i % world).  With per-item weights (e.g.
    segment lengths): greedy longest-first assignment to the least-loaded rank, which every
    rank computes identically (deterministic tie-breaks), so no communication is needed."""
    if weights is None:
        return list(range(rank, n_items, world))
    order = sorted(range(n_items), key=lambda i: (-float(weights[i]), i))
    load = [0.0] * world
    mine = []
    for i in order:
        r = min(range(world), key=lambda q: (load[q], q))
        load[r] += float(weights[i])
        if r == rank:
            mine.append(i)
    return sorted(mine)


def broadcast_module(module: torch.nn.Module, src: int = 0):
    """Rank `src`'s parameters and buffers overwrite everyone else's (start-up only)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)


def allreduce_logmel_stats(stats: torch.Tensor) -> torch.Tensor:
    """stats = [sum, count] (float64) of this rank's part of ONE reference call; returns the
    global mean as a float32 scalar tensor, identical on every rank."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    return (stats[0] / stats[1]).to(torch.float32).reshape(1)


def gather_segments(local: dict, n_items: int, dst: int = 0, like=None):
    """Gather variable-length per-segment results to rank `dst`.

    local: {item_index: tensor [T_i, ...]} for the items this rank processed (same trailing
    shape and dtype everywhere).  Returns the list of n_items tensors (in item order) on
    `dst`, None elsewhere.  Protocol: all_gather of (index, length) pairs, then one padded
    all_gather-free exchange per rank pair via dist.gather of a flat buffer.
    A rank that processed nothing (more ranks than items) passes `like`, any tensor with the results' trailing
    shape, dtype and device; without it the description is fetched from a rank that holds data (one small
    all_gather_object)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
        return [local[i] for i in range(n_items)]
    items = sorted(local)
    sample = local[items[0]] if items else like
    if like is None:
        # every rank must agree on whether the description exchange happens: it does whenever n_items < world
        # (some rank is necessarily empty) -- a pure function of arguments all ranks share
        if n_items < world:
            desc = [None] * world
            mine = None if sample is None else (tuple(sample.shape[1:]), str(sample.dtype).replace("torch.", ""), str(sample.device.type))
            dist.all_gather_object(desc, mine)
            if sample is None:
                got = next((d for d in desc if d is not None), None)
                if got is None:
                    raise RuntimeError("gather_segments: no rank holds any item")
                dev_type = got[2]
                device = torch.device("cuda", torch.cuda.current_device()) if dev_type == "cuda" else torch.device("cpu")
                sample = torch.empty((0,) + got[0], dtype=getattr(torch, got[1]), device=device)
    if sample is None:
        raise RuntimeError("gather_segments: this rank holds no item; pass `like`")
    # 1) who holds what, and how long
    meta = torch.full((n_items, 2), -1, dtype=torch.int64)
    for i in items:
        meta[i, 0] = rank
        meta[i, 1] = local[i].shape[0]
    dev = sample.device
    meta = meta.to(dev)
    dist.all_reduce(meta, op=dist.ReduceOp.MAX)
    meta = meta.cpu()
    if int(meta[:, 0].min()) < 0:
        raise RuntimeError("gather_segments: some item was processed by no rank")
    # trailing shape / dtype from any rank that holds data (every rank holds >= 1 item in practice)
    trailing = tuple(sample.shape[1:])
    per_rank_rows = [int(meta[meta[:, 0] == r, 1].sum()) for r in range(world)]
    flat = torch.cat([local[i] for i in items], dim=0) if items else torch.empty((0,) + trailing, dtype=sample.dtype, device=dev)
    pad = max(per_rank_rows)
    buf = torch.zeros((pad,) + trailing, dtype=flat.dtype, device=dev)
    buf[: flat.shape[0]] = flat
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    result = [None] * n_items
    cursor = [0] * world
    for i in range(n_items):
        r, n = int(meta[i, 0]), int(meta[i, 1])
        result[i] = out[r][cursor[r]: cursor[r] + n]
        cursor[r] += n
    return result


def encode_clip_sharded(encoder, mel: torch.Tensor, out_tile: int, dst: int = 0, batch: int = 8):
    """ONE clip over all ranks ("single hour-long clip across GPUs"): every rank holds the clip's
    log-mel [1, T, C] (0.9 ms per hour of audio: not worth sharding, and its mean needs the whole clip anyway), encodes
    the tiles `rank::world` of tiling.plan_tiles with their [-640, +780]-frame halo, and rank `dst` receives the
    stitched encoder output [1, T', C'] (None elsewhere).  No collective but the final gather."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    plan = tiling.plan_tiles(int(mel.shape[1]), out_tile, tuple(encoder.depths))
    if not plan:
        return torch.zeros(1, 0, encoder.sizes[-1], dtype=torch.float32, device=mel.device) if rank == dst else None
    mine = tiling.shard_tiles(plan, rank, world)
    done = tiling.encode_tiles(encoder, mel, mine, batch)
    local = {plan.index(t): y.contiguous() for t, y in done.items()}
    like = torch.empty(0, encoder.sizes[-1], dtype=torch.float32, device=mel.device)
    parts = gather_segments(local, len(plan), dst=dst, like=like)
    return torch.cat(parts, dim=0)[None] if parts is not None else None
