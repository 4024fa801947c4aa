"""Device side of the hypothesis-embedding step that feeds WDER scoring
(tal/utils/aligned_to_wder_format.py:150-214): for every generated
token, pool the SDModel features of its cross-attention window with the attention weights that
`generate_unaligned` recorded, and vote a speaker id from the per-frame arg-max ids.

The text side of that script (word segmentation through the sentencepiece model, role maps,
pickles) is host plumbing that needs files the reference does not ship; it is not built.
"""
import torch

from . import _native as N
from . import ops


def pool_attention_features(attention, chunk_start, features):
    """attention [N, S] (rows as returned by generate_unaligned), chunk_start [N] (int),
    features [T', E] (SDModel.spk_embed_proj output of the episode) -> [N, E]:
    emb[n] = attention[n, :len] @ features[cs:cs+S]  (aligned_to_wder_format.py:203-213)."""
    lib = N.lib()
    attention = ops._f32c(attention, "pool_attention_features(attention)")
    features = ops._f32c(features, "pool_attention_features(features)")
    cs = chunk_start.to(device=attention.device, dtype=torch.int64).contiguous()
    n, s = attention.shape
    t, e = features.shape
    out = torch.empty(n, e, dtype=torch.float32, device=attention.device)
    N.check(lib.tal_attn_pool_fwd(N.ptr(attention), N.ptr(cs), N.ptr(features), t, e, n, s, N.ptr(out),
                                  N.stream_handle()), "tal_attn_pool_fwd")
    return out


def vote_speaker_ids(attention, chunk_start, frame_ids):
    """Most heavily attended speaker id per token (aligned_to_wder_format.py:158-166,194-196):
    returns (ids [N] int32, weight [N])."""
    lib = N.lib()
    attention = ops._f32c(attention, "vote_speaker_ids(attention)")
    cs = chunk_start.to(device=attention.device, dtype=torch.int64).contiguous()
    ids = frame_ids.to(device=attention.device, dtype=torch.int32).contiguous()
    n, s = attention.shape
    out = torch.empty(n, dtype=torch.int32, device=attention.device)
    wgt = torch.empty(n, dtype=torch.float32, device=attention.device)
    N.check(lib.tal_attn_vote_fwd(N.ptr(attention), N.ptr(cs), N.ptr(ids), ids.numel(), n, s, N.ptr(out), N.ptr(wgt),
                                  N.stream_handle()), "tal_attn_vote_fwd")
    return out, wgt
