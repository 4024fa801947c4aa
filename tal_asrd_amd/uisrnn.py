"""Host-side mirror of UIS-RNN's CoreRNN (tal/diarization/uisrnn/uisrnn.py:20-39) on the
tal_gru_cell_fwd / tal_linear_fwd kernels.  Same constructor, forward signature and
state_dict keys (gru.weight_ih_l0 ..., linear_mean1.*, linear_mean2.*).  The CRP beam
search around it (uisrnn.py:395-554) is host control flow and out of scope."""
import torch
import torch.nn as nn

from . import _native as N
from . import ops
from .models import Linear


class CoreRNN(nn.Module):
    def __init__(self, input_dim, hidden_size, depth, observation_dim, dropout=0):
        super().__init__()
        self.hidden_size = hidden_size
        self.depth = depth
        if depth >= 2:
            self.gru = nn.GRU(input_dim, hidden_size, depth, dropout=dropout)
        else:
            self.gru = nn.GRU(input_dim, hidden_size, depth)
        self.linear_mean1 = Linear(hidden_size, hidden_size)
        self.linear_mean2 = Linear(hidden_size, observation_dim)
        self.eval()

    def _cell(self, layer, x, h):
        lib = N.lib()
        g = self.gru
        w_ih, w_hh = getattr(g, "weight_ih_l%d" % layer), getattr(g, "weight_hh_l%d" % layer)
        b_ih, b_hh = getattr(g, "bias_ih_l%d" % layer), getattr(g, "bias_hh_l%d" % layer)
        B, In = x.shape
        H = self.hidden_size
        out = torch.empty(B, H, dtype=torch.float32, device=x.device)
        nws = lib.tal_gru_cell_workspace_bytes(B, H)
        ws = ops._ws(nws, x.device)
        N.check(lib.tal_gru_cell_fwd(N.ptr(x), N.ptr(h), B, In, H, N.ptr(w_ih), N.ptr(w_hh), N.ptr(b_ih), N.ptr(b_hh),
                                     N.ptr(out), N.ptr(ws), nws, N.stream_handle()), "tal_gru_cell_fwd")
        return out

    @torch.no_grad()
    def forward(self, input_seq, hidden=None):
        """input_seq [L, B, In], hidden [depth, B, H] | None -> (mean [L, B, obs], hidden [depth, B, H])."""
        if isinstance(input_seq, torch.nn.utils.rnn.PackedSequence):
            raise N.NativeError("CoreRNN: packed sequences are a training-time path (uisrnn.py:35-37), not built")
        x = ops._f32c(input_seq, "CoreRNN.forward")
        L, B, _ = x.shape
        if hidden is None:
            hs = [torch.zeros(B, self.hidden_size, dtype=torch.float32, device=x.device) for _ in range(self.depth)]
        else:
            hidden = ops._f32c(hidden, "CoreRNN.forward(hidden)")
            hs = [hidden[l].contiguous() for l in range(self.depth)]
        outs = []
        for t in range(L):
            inp = x[t].contiguous()
            for l in range(self.depth):
                hs[l] = self._cell(l, inp, hs[l])
                inp = hs[l]
            outs.append(inp)
        out = torch.stack(outs, 0)
        m1 = ops.linear(out, self.linear_mean1.weight, self.linear_mean1.bias, mode=1)
        mean = ops.linear(m1, self.linear_mean2.weight, self.linear_mean2.bias)
        return mean, torch.stack(hs, 0)
