"""Time tiling of the TDS encoder (SURVEY §8b `halo_mode`, §8e "single hour-long clip across GPUs").

The encoder (tal/asr/models.py:349-397) is a stack of stride-2 k=21 convs without padding and TDSBlocks whose k=21 conv
pads 10 zeros at the TRUE ends of the sequence (models.py:304-308).  Output frame t therefore reads the mel frames
[8 t - left, 8 t + right] (left = 640, right = 780 for the 2 / 3 / 6 block stack), and a slice of the sequence that
carries this halo reproduces the frames whose window lies inside it exactly as the whole sequence would -- the frames at
a true end need no halo, their zero padding is the same in the slice (slices start at multiples of the total stride, so
every stage's index 0 of the slice is a stage index of the whole).

What it is for: items longer than the 2 GiB-per-item limit of the fp16x3 kernels (~3.7 h of audio), and one clip spread
over several GPUs (`tiles[rank::world]`, results through distributed.gather_segments).  One GPU holds an hour of audio a
hundred times over, so nothing on the bench path tiles.
"""
from dataclasses import dataclass
from typing import List, Sequence, Tuple

import torch

KERNEL_SIZE = 21


def tds_out_len(T: int, n_stages: int = 3) -> int:
    """Frames after the stride-2, k=21, padding-0 convs (models.py:363-364)."""
    for _ in range(n_stages):
        if T < KERNEL_SIZE:
            return 0
        T = (T - KERNEL_SIZE) // 2 + 1
    return T


def receptive_halo(depths: Sequence[int], kernel_size: int = KERNEL_SIZE) -> Tuple[int, int, int]:
    """(left, right, stride): output frame t of the stack reads input frames [stride t - left, stride t + right]."""
    half = kernel_size // 2
    lo, hi, stride = 0, 0, 1            # window of one frame at the current resolution, in units of that resolution
    for d in reversed(list(depths)):
        # d TDSBlocks (each +-half at this resolution), below them the stride-2 conv: frame u reads [2 u, 2 u + k - 1]
        lo, hi = lo - d * half, hi + d * half
        lo, hi = 2 * lo, 2 * hi + kernel_size - 1
        stride *= 2
    return -lo, hi, stride


@dataclass(frozen=True)
class Tile:
    in_start: int      # first input (mel) frame of the slice
    in_stop: int       # one past its last
    out_start: int     # first output frame this tile is responsible for (global index)
    out_stop: int      # one past the last
    skip: int          # output frames of the slice's own result that precede out_start


def plan_tiles(T: int, out_tile: int, depths: Sequence[int] = (2, 3, 6)) -> List[Tile]:
    """Tiles of `out_tile` output frames over an input of T frames; [] if the input is shorter than one output frame."""
    if out_tile < 1:
        raise ValueError("plan_tiles: out_tile must be positive")
    left, right, stride = receptive_halo(depths)
    t_out = tds_out_len(T, len(list(depths)))
    tiles = []
    for o0 in range(0, t_out, out_tile):
        o1 = min(o0 + out_tile, t_out)
        a = max(stride * (o0 - -(-left // stride)), 0)  # the halo rounded up to the stride: every stage's index 0 of the
                                                       # slice is a stage index of the whole sequence
        b = min(stride * (o1 - 1) + right + 1, T)
        tiles.append(Tile(a, b, o0, o1, o0 - a // stride))
    return tiles


def max_item_frames(sizes: Sequence[int] = (80, 800, 1120, 1440)) -> int:
    """Input frames one call may carry before some stage's activation ([T / 2^s, sizes[s]] floats) passes the 2 GiB per-item
    limit of the fp16x3 kernels' 32-bit buffer offsets (the 2x model: stage 1, [T / 2, 800] -> ~3.7 h of audio)."""
    limit = None
    for s, c in enumerate(sizes):
        rows = (1 << 31) // (4 * int(c)) - 64           # rows of this stage that fit, with a margin for the conv halo
        frames = rows << s
        limit = frames if limit is None else min(limit, frames)
    return int(limit)


def encode_tiles(encoder, mel: torch.Tensor, tiles: Sequence[Tile], batch: int = 8) -> dict:
    """{tile: its output frames [n, C']} for some tiles of a plan over mel [1, T, C] (time-major).  Equal-length slices go
    through the encoder `batch` at a time: a batch item's ends are true ends for its zero padding, which is exactly what
    the halo absorbs."""
    if mel.dim() != 3 or mel.shape[0] != 1:
        raise ValueError("encode_tiles: mel must be [1, T, C]")
    by_len = {}
    for t in tiles:
        by_len.setdefault(t.in_stop - t.in_start, []).append(t)
    out = {}
    for group in by_len.values():
        for i in range(0, len(group), batch):
            chunk = group[i:i + batch]
            x = torch.stack([mel[0, t.in_start:t.in_stop] for t in chunk])
            y = encoder.forward_time_major(x.contiguous())
            for j, t in enumerate(chunk):
                out[t] = y[j, t.skip:t.skip + (t.out_stop - t.out_start)]
    return out


def encode_tiled(encoder, mel: torch.Tensor, out_tile: int, tiles: Sequence[Tile] = None, batch: int = 8) -> torch.Tensor:
    """encoder.forward_time_major(mel) computed tile by tile.  mel [1, T, C] (time-major) -> [1, T', C'].
    `tiles` restricts the work to some tiles of the plan (a rank's share); the frames of the others stay zero."""
    if mel.dim() != 3 or mel.shape[0] != 1:
        raise ValueError("encode_tiled: mel must be [1, T, C]")
    depths = tuple(encoder.depths)
    T = int(mel.shape[1])
    todo = plan_tiles(T, out_tile, depths) if tiles is None else list(tiles)
    out = torch.zeros(1, tds_out_len(T, len(depths)), encoder.sizes[-1], dtype=torch.float32, device=mel.device)
    for t, y in encode_tiles(encoder, mel, todo, batch).items():
        out[0, t.out_start:t.out_stop] = y
    return out


def shard_tiles(plan: Sequence[Tile], rank: int, world: int) -> List[Tile]:
    """A rank's tiles of one clip: round-robin (tiles cost the same but for the last)."""
    return list(plan[rank::world])
