"""Host-side mirror of tal/modules.py (reference): PositionalEncoding, weight_init.

`PositionalEncoding` keeps the reference's constructor and its `pe` buffer (part
of reference checkpoints as `pos_dec_encoder.pe`); the add runs in
the token-embedding HIP kernel when called through ASRModel.decode, and through
`tal_add_rows` when the module is called on its own.
"""
import math

import torch
import torch.nn as nn

from . import _native as N


def sinusoid_table(max_len: int, d_model: int) -> torch.Tensor:
    """pe[pos, 2i] = sin(pos / 10000^(2i/d)), pe[pos, 2i+1] = cos(...) (tal/modules.py:45-50),
    evaluated with the same float32 torch ops so the buffer is bit-identical."""
    pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    inv = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    table = torch.zeros(max_len, d_model)
    table[:, 0::2] = torch.sin(pos * inv)
    table[:, 1::2] = torch.cos(pos * inv)
    return table


class PositionalEncoding(nn.Module):
    """x [B, U, d] -> x + pe[:U] (tal/modules.py:41-64).  Dropout is identity at inference."""

    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        self.register_buffer("pe", sinusoid_table(max_len, d_model))

    def forward(self, x):
        N.require_cuda(x, "PositionalEncoding.forward")
        if x.size(1) > self.pe.size(0):
            raise IndexError("sequence length %d exceeds max_len %d" % (x.size(1), self.pe.size(0)))
        from .ops import add_positional
        return add_positional(x, self.pe)


def weight_init():
    """tal/modules.py:6-21: reset_parameters() everywhere, embeddings U(+-1/sqrt(dim))."""
    def apply_to(module):
        reset = getattr(module, "reset_parameters", None)
        if callable(reset):
            reset()
        if isinstance(module, nn.Embedding):
            bound = 1.0 / math.sqrt(module.weight.size(1))
            module.weight.data.uniform_(-bound, bound)
    return apply_to
