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
    """Rank `src`'s parameters and buffers overwrite everyone else's (start-up only): ONE broadcast per dtype of a flat
    buffer (SDModel: 41 M floats = 165 MB in one collective instead of ~150 small ones)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    by_dtype = {}
    for t in list(module.parameters()) + list(module.buffers()):
        by_dtype.setdefault((t.dtype, t.device), []).append(t.data)
    for (dtype, device), tensors in sorted(by_dtype.items(), key=lambda kv: str(kv[0])):
        flat = torch.cat([t.reshape(-1) for t in tensors]) if dist.get_rank() == src else \
            torch.empty(sum(t.numel() for t in tensors), dtype=dtype, device=device)
        dist.broadcast(flat, src=src)
        if dist.get_rank() != src:
            o = 0
            for t in tensors:
                t.copy_(flat[o:o + t.numel()].view_as(t))
                o += t.numel()


def allreduce_logmel_stats(stats: torch.Tensor) -> torch.Tensor:
    """stats = [sum, count] (float64) of this rank's part of ONE reference call; returns the
    global mean as a float32 scalar tensor, identical on every rank."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    return (stats[0] / stats[1]).to(torch.float32).reshape(1)


_DTYPES = [torch.float32, torch.float64, torch.float16, torch.bfloat16, torch.int32, torch.int64, torch.int16, torch.int8,
           torch.uint8, torch.bool]


class _DoneWork:
    """What SegmentGather.gather(async_op=True) returns as `work` when there is nothing to wait for (one rank)."""

    def wait(self, timeout=None):
        return True

    def is_completed(self):
        return True


class SegmentGather:
    """Plan + buffers for gathering variable-length per-segment results to rank `dst`, built ONCE for a fixed assignment
    (which rank holds which item, how many rows each has): the (owner, rows) table and the row description travel in one
    small all-reduce here, so `gather` is exactly one collective per call, with no host synchronisation of its own.

    Every decision is derived from data all ranks share after that all-reduce -- a rank without items needs no `like`
    argument, and an inconsistent call (an item nobody holds, two owners, differing row shapes) raises on EVERY rank
    instead of leaving some of them blocked in a collective.

    lengths: {item_index: rows} of the items THIS rank holds; trailing / dtype: shape and dtype of a row (None on a rank
    that holds nothing)."""

    def __init__(self, lengths: dict, n_items: int, trailing=None, dtype=None, device=None, dst: int = 0):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.n_items, self.dst = int(n_items), int(dst)
        self.device = torch.device("cpu") if device is None else torch.device(device)
        self.local_items = sorted(lengths)
        if self.world == 1:
            self.trailing, self.dtype = tuple(trailing or ()), dtype
            return
        # table[i] = (owner, rows) of item i, then one description row: (ndim, d0..d5, dtype code); -1 = "not mine"
        table = torch.full((n_items + 1, 8), -1, dtype=torch.int64)
        for i, n in lengths.items():
            if not 0 <= i < n_items:
                raise ValueError("SegmentGather: item index %d outside [0, %d)" % (i, n_items))
            table[i, 0], table[i, 1] = self.rank, int(n)
        if lengths:
            if trailing is None or dtype is None:
                raise ValueError("SegmentGather: a rank that holds items must describe their rows (trailing, dtype)")
            if len(trailing) > 6:
                raise ValueError("SegmentGather: rows of more than 6 dimensions")
            table[n_items, 0] = len(trailing)
            for k, d in enumerate(trailing):
                table[n_items, 1 + k] = int(d)
            table[n_items, 7] = _DTYPES.index(dtype)
        lo = table.clone()
        lo[lo < 0] = torch.iinfo(torch.int64).max              # MIN over ranks that do hold / describe the entry
        both = torch.stack([table, -lo]).to(self.device)       # one MAX all-reduce yields max and -min
        dist.all_reduce(both, op=dist.ReduceOp.MAX)
        both = both.cpu()
        hi, lo = both[0], -both[1]
        problems = []
        if int(hi[:n_items, 0].min()) < 0:
            problems.append("items %s were processed by no rank" % [i for i in range(n_items) if int(hi[i, 0]) < 0][:8])
        differ = (hi != lo) & (hi >= 0)          # an entry some rank filled in, with different values on different ranks
        if bool(differ[:n_items].any()):
            problems.append("some item is claimed by two ranks")
        if int(hi[n_items, 0]) < 0:
            problems.append("no rank holds any item")
        elif bool(differ[n_items].any()):
            problems.append("ranks disagree on the row shape / dtype")
        if problems:       # (identical on every rank: all of them raise, none is left inside a collective)
            raise RuntimeError("SegmentGather: " + "; ".join(problems))
        self.owner = [int(v) for v in hi[:n_items, 0]]
        self.length = [int(v) for v in hi[:n_items, 1]]
        self.trailing = tuple(int(v) for v in hi[n_items, 1:1 + int(hi[n_items, 0])])
        self.dtype = _DTYPES[int(hi[n_items, 7])]
        rows = [0] * self.world
        for o, n in zip(self.owner, self.length):
            rows[o] += n
        self.rows_per_rank = rows
        pad = max(rows)
        self.send = torch.zeros((pad,) + self.trailing, dtype=self.dtype, device=self.device)
        self.recv = [torch.empty_like(self.send) for _ in range(self.world)] if self.rank == dst else None

    def gather(self, local: dict, async_op: bool = False):
        """local: {item: tensor [rows_i, *trailing]} of this rank's items -> list of n_items tensors (views into the plan's
        receive buffers, item order) on `dst`, None elsewhere.  async_op: returns (work, finish) instead; call
        work.wait() and then finish() for the list.  The plan owns ONE send buffer: an asynchronous gather must be waited
        for before the next gather of the same plan is issued (asserted)."""
        if self.world == 1:
            items = [local[i] for i in range(self.n_items)]
            return (_DoneWork(), lambda: items) if async_op else items
        if getattr(self, "_in_flight", None) is not None and not self._in_flight.is_completed():
            raise RuntimeError("SegmentGather.gather: the previous asynchronous gather of this plan has not been waited for "
                               "(its send buffer is still in flight)")
        o = 0
        for i in self.local_items:
            n = self.length[i]
            if tuple(local[i].shape) != (n,) + self.trailing:
                raise ValueError("SegmentGather.gather: item %d has shape %s, planned %s" % (i, tuple(local[i].shape), (n,) + self.trailing))
            self.send[o:o + n].copy_(local[i])
            o += n
        work = dist.gather(self.send, self.recv, dst=self.dst, async_op=async_op)
        self._in_flight = work if async_op else None

        def finish():
            if self.rank != self.dst:
                return None
            cursor = [0] * self.world
            out = [None] * self.n_items
            for i in range(self.n_items):
                r, n = self.owner[i], self.length[i]
                out[i] = self.recv[r][cursor[r]:cursor[r] + n]
                cursor[r] += n
            return out
        return (work, finish) if async_op else finish()


def pack_feat_ids(feat: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """[T, E] float32 features + [T] int32 ids -> [T, E + 1] float32 with the ids bit-cast into the last column, so that
    both leave in ONE collective (SegmentGather.gather); unpack_feat_ids undoes it."""
    out = torch.empty(feat.shape[0], feat.shape[1] + 1, dtype=torch.float32, device=feat.device)
    out[:, :-1] = feat
    out[:, -1] = ids.to(torch.int32).view(torch.float32)
    return out


def unpack_feat_ids(packed: torch.Tensor):
    return packed[:, :-1], packed[:, -1].contiguous().view(torch.int32)


def gather_segments(local: dict, n_items: int, dst: int = 0, like=None):
    """Gather variable-length per-segment results to rank `dst`: one-shot form of SegmentGather (plan + gather).

    local: {item_index: tensor [T_i, ...]} for the items this rank processed (same trailing shape and dtype everywhere).
    Returns the list of n_items tensors (in item order) on `dst`, None elsewhere.  `like` (any tensor with the results'
    device) is only used for its device on a rank that holds nothing; every rank runs the same collectives whatever it
    passes."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return [local[i] for i in range(n_items)]
    items = sorted(local)
    sample = local[items[0]] if items else like
    if sample is not None:
        device = sample.device
    else:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    plan = SegmentGather({i: int(local[i].shape[0]) for i in items}, n_items,
                         trailing=tuple(local[items[0]].shape[1:]) if items else None,
                         dtype=local[items[0]].dtype if items else None, device=device, dst=dst)
    return plan.gather(local)


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
