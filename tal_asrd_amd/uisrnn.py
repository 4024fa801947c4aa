"""Host-side mirror of UIS-RNN's CoreRNN (tal/diarization/uisrnn/uisrnn.py:20-39) on the
tal_gru_cell_fwd / tal_linear_fwd kernels.  Same constructor, forward signature and
state_dict keys (gru.weight_ih_l0 ..., linear_mean1.*, linear_mean2.*), and the inference
side of `UISRNN` (uisrnn.py:64-87,136-151,371-583): the CRP beam search over speaker
assignments, restated so that the device sees ONE batched GRU step per observation (all
surviving beams at once) instead of one launch-bound B=1 GRU call per (beam, candidate
cluster) -- see `UISRNN.predict_single`.  Training (`fit*`, uisrnn.py:157-369) is out of scope."""
import numpy as np
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


class _Beam:
    """uisrnn.py:40-62.  means: host float32 [D] per cluster; hidden: device [depth, 1, H] per cluster."""
    __slots__ = ("means", "hidden", "trace", "block_counts", "nll")

    def __init__(self, src=None):
        if src is None:
            self.means, self.hidden, self.trace, self.block_counts, self.nll = [], [], [], [], 0
        else:
            self.means, self.hidden = list(src.means), list(src.hidden)
            self.trace, self.block_counts, self.nll = list(src.trace), list(src.block_counts), src.nll


class UISRNN:
    """Inference half of the reference's UISRNN (uisrnn.py:64-87 constructor fields, :136-151 load,
    :371-583 beam search).  `args` objects are duck-typed like the reference's argparse namespaces
    (observation_dim, rnn_hidden_size, rnn_depth, rnn_dropout, sigma2, transition_bias, crp_alpha;
    inference: beam_size, look_ahead, test_iteration)."""

    def __init__(self, args, rnn_model=None, device=None):
        self.observation_dim = args.observation_dim
        self.device = torch.device(device if device is not None else "cuda:0")
        self.rnn_model = rnn_model if rnn_model is not None else CoreRNN(
            self.observation_dim, args.rnn_hidden_size, args.rnn_depth, self.observation_dim,
            getattr(args, "rnn_dropout", 0)).to(self.device)
        self.rnn_init_hidden = torch.zeros(args.rnn_depth, 1, args.rnn_hidden_size, device=self.device)
        sigma2 = 0.1 if getattr(args, "sigma2", None) is None else args.sigma2      # _INITIAL_SIGMA2_VALUE
        self.sigma2 = sigma2 * torch.ones(self.observation_dim)
        self.transition_bias = getattr(args, "transition_bias", None)
        self.transition_bias_denominator = 0.0
        self.crp_alpha = getattr(args, "crp_alpha", 1.0)

    def load(self, filepath):
        var = torch.load(filepath, map_location="cpu", weights_only=False)
        self.load_dict(var)

    def load_dict(self, var):
        """The checkpoint dict of uisrnn.py:123-134."""
        self.rnn_model.load_state_dict(var["rnn_state_dict"])
        self.rnn_model.to(self.device)
        self.rnn_init_hidden = torch.as_tensor(np.asarray(var["rnn_init_hidden"]), dtype=torch.float32).to(self.device)
        self.transition_bias = float(var["transition_bias"])
        self.transition_bias_denominator = float(var["transition_bias_denominator"])
        self.crp_alpha = float(var["crp_alpha"])
        self.sigma2 = torch.as_tensor(np.asarray(var["sigma2"]), dtype=torch.float32)

    # -- arithmetic of one candidate's loss, on the host like the reference (.cpu().detach().numpy(), :404,433)
    def _mse(self, mean, x, w):
        """loss_func.weighted_mse_loss for one [D] pair: mean over D of diff^2 * w, times D / (#rows whose
        first entry differs) -- a single row here, so a division by 1, or by 0 (inf/nan) when the first
        entries agree exactly (kept)."""
        d2 = (mean - x) ** 2
        nz = np.float32(1.0 if d2[0] != 0 else 0.0)
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.float32(np.mean(d2 * w, dtype=np.float32) * np.float32(d2.size) * np.float32(1.0) / nz)

    def _candidate_loss(self, beam, c, x, w, new_mean):
        """Negative log-likelihood increment of assigning observation x to cluster c (uisrnn.py:399-440):
        float32 MSE term, the CRP / transition terms subtracted in float64 and rounded back to float32
        (the reference's in-place `-=` on a float32 0-d array)."""
        if c < len(beam.means):
            loss = self._mse(beam.means[c], x, w)
            if c == beam.trace[-1]:
                adj = np.log(1 - self.transition_bias)
            else:
                adj = (np.log(self.transition_bias) + np.log(beam.block_counts[c])
                       - np.log(sum(beam.block_counts) + self.crp_alpha))
        else:
            loss = self._mse(new_mean, x, w)
            adj = (np.log(self.transition_bias) + np.log(self.crp_alpha)
                   - np.log(sum(beam.block_counts) + self.crp_alpha))
        return np.float32(np.float64(loss) - adj)

    def _apply(self, picks, x_dev, new_hidden):
        """Advance the picked (beam, cluster, loss) candidates by one observation: ONE batched GRU step +
        mean head for all of them, one D2H copy of the means, then the host bookkeeping of
        uisrnn.py:412-441 (running cluster mean with the reference's pre-append count, block counts, trace)."""
        R = len(picks)
        h_in = torch.cat([(b.hidden[c] if c < len(b.means) else new_hidden) for b, c, _ in picks], dim=1)
        mean, h_out = self.rnn_model(x_dev.view(1, 1, -1).expand(1, R, -1).contiguous(), h_in.contiguous())
        mean = mean[0].cpu().numpy()
        out = []
        for r, (b, c, loss) in enumerate(picks):
            nb = _Beam(b)
            hid = h_out[:, r:r + 1]
            if c < len(nb.means):
                last = nb.trace[-1]
                n = float((np.asarray(nb.trace) == c).sum())        # count BEFORE this observation is appended
                with np.errstate(divide="ignore", invalid="ignore"):
                    nb.means[c] = ((nb.means[c] * np.float32(n - 1.0) + mean[r]) / np.float32(n)).astype(np.float32)
                nb.hidden[c] = hid
                if c != last:
                    nb.block_counts[c] += 1
                nb.trace.append(c)
            else:
                nb.means.append(mean[r].copy())
                nb.hidden.append(hid)
                nb.block_counts.append(1)
                nb.trace.append(c)
            nb.nll = nb.nll + loss
            out.append(nb)
        return out

    @torch.no_grad()
    def predict_single(self, test_sequence, args):
        """uisrnn.py:470-554.  Same checks, same ranking (flattened argsort over the
        [beam, clusters+1, clusters+2, ...] score array), same returned trace.

        Device schedule (the MI355X-first part): the reference calls the B=1 GRU for EVERY (beam,
        candidate cluster) to score it and again for the winners -- O(N * beam * clusters) launch-bound
        GEMVs.  A candidate's score only needs the cluster means, which are already on the host, so here
        the scores of the last look-ahead position are computed without touching the GPU and only the
        <= beam_size winners are advanced, all in one batched tal_gru_cell_fwd + mean-head call.  The
        new-cluster prior (GRU of a zero observation from rnn_init_hidden) is a constant computed once.
        For look_ahead > 1 the inner positions are expanded level by level, one batched call per level."""
        if not isinstance(test_sequence, np.ndarray) or test_sequence.dtype != float:
            raise TypeError("test_sequence should be a numpy array of float type.")
        if test_sequence.ndim != 2:
            raise ValueError("test_sequence must be 2-dim array.")
        n_obs, dim = test_sequence.shape
        if dim != self.observation_dim:
            raise ValueError("test_sequence does not match the dimension specified by args.observation_dim.")
        beam_size, look_ahead, iters = int(args.beam_size), int(args.look_ahead), int(args.test_iteration)
        seq = np.tile(test_sequence, (iters, 1)).astype(np.float32)
        seq_dev = torch.from_numpy(seq).to(self.device)
        w = (1.0 / (2.0 * self.sigma2.detach().cpu().numpy().astype(np.float32))).astype(np.float32)
        zero = torch.zeros(1, 1, dim, dtype=torch.float32, device=self.device)
        m0, new_hidden = self.rnn_model(zero, self.rnn_init_hidden.to(self.device).float().contiguous())
        new_mean = m0.reshape(-1).cpu().numpy()

        beams = [_Beam()]
        for t0 in range(0, iters * n_obs, look_ahead):
            la = min(look_ahead, iters * n_obs - t0)
            kmax = max(len(b.means) for b in beams)
            shape = [beam_size] + [kmax + 1 + j for j in range(la)]
            scores = np.full(shape, np.inf)
            # frontier: (flat index prefix, root beam rank, state, accumulated nll, pick history)
            frontier = [((r,), b, []) for r, b in enumerate(beams)]
            for j in range(la):
                x = seq[t0 + j]
                cands = []
                for idx, b, hist in frontier:
                    for c in range(len(b.means) + 1):
                        cands.append((idx + (c,), b, c, self._candidate_loss(b, c, x, w, new_mean), hist))
                if j == la - 1:
                    for idx, b, c, loss, hist in cands:
                        scores[idx] = b.nll + loss
                    last = cands
                else:
                    adv = self._apply([(b, c, loss) for _, b, c, loss, _ in cands], seq_dev[t0 + j], new_hidden)
                    frontier = [(idx, nb, hist + [(c, loss)]) for (idx, _, c, loss, hist), nb in zip(cands, adv)]
            flat = scores.reshape(-1)
            order = np.argsort(flat, axis=None)
            ranked = np.sort(flat, axis=None)
            ranked[ranked == np.inf] = 0
            n_new = min(len(np.trim_zeros(ranked)), beam_size)
            by_idx = {idx: (b, c, loss, hist) for idx, b, c, loss, hist in last}
            if la == 1:
                picks = []
                for k in range(n_new):
                    idx = tuple(int(v) for v in np.unravel_index(order[k], scores.shape))
                    b, c, loss, _ = by_idx[idx]
                    picks.append((b, c, loss))
                beams = self._apply(picks, seq_dev[t0], new_hidden)
            else:
                # winners restart from their root beam and replay their cluster sequence, batched per position
                roots, seqs_c = [], []
                for k in range(n_new):
                    idx = tuple(int(v) for v in np.unravel_index(order[k], scores.shape))
                    _, c, loss, hist = by_idx[idx]
                    roots.append(beams[idx[0]])
                    seqs_c.append(hist + [(c, loss)])
                cur = roots
                for j in range(la):
                    cur = self._apply([(cur[k], seqs_c[k][j][0], seqs_c[k][j][1]) for k in range(n_new)],
                                      seq_dev[t0 + j], new_hidden)
                beams = cur
        return beams[0].trace[-n_obs:]

    def predict(self, test_sequences, args, debug=False):
        """uisrnn.py:556-583."""
        if isinstance(test_sequences, np.ndarray):
            return self.predict_single(test_sequences, args)
        if isinstance(test_sequences, list):
            return [self.predict_single(s, args) for s in test_sequences]
        raise TypeError("test_sequences should be either a list or numpy array.")
