"""A stand-in language model for the shallow-fusion branch of System.generate (tal/asr/system.py:127-138).  The reference's own LM
class does not exist in its tree (`tal/lm` is absent), so the branch is pinned with a caller-side model both sides
can run: tokens [rows, U] -> logits [rows, U, vocab], called as lm(tokens, causal_mask=False) like the reference does.  Test
infrastructure only (the recording script and the GPU test build the same module from the same seed)."""
import torch
from torch import nn


class StandInLM(nn.Module):
    def __init__(self, vocab=10000, width=48, seed=2024):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.emb = nn.Embedding(vocab, width)
        self.out = nn.Linear(width, vocab)
        with torch.no_grad():
            self.emb.weight.copy_(torch.randn(vocab, width, generator=g))
            self.out.weight.copy_(torch.randn(vocab, width, generator=g) * 0.6)
            self.out.bias.copy_(torch.randn(vocab, generator=g) * 0.5)

    def forward(self, tokens, causal_mask=True):
        # a running mean of the prefix embeddings (causal or not makes no difference to the LAST position, the only one read)
        e = self.emb(tokens)
        h = torch.cumsum(e, dim=1) / torch.arange(1, e.shape[1] + 1, device=e.device, dtype=e.dtype).view(1, -1, 1)
        return self.out(torch.tanh(h))
