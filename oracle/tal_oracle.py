"""CPU oracle for the tal-asrd acoustic hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package (tal_asrd_amd) never does, and its HIP path
fails loudly when the native library is missing instead of falling back here.

What it is: a functional (stateless) fp32 restatement, on PyTorch-CPU ops, of
the reference's acoustic path.  Weights come in as a flat dict keyed by the
reference's own state_dict names.  Each function cites the reference lines it
follows (paths relative to /root/reference).

Pinning status
--------------
* TDS encoder, heads, padding mask, positional encoding, decoder layers,
  decode / decode_spk, CoreRNN: PINNED by golden vectors produced by importing
  the reference's own modules in the build container
  (tests/golden/make_golden.py -> tests/golden/*.npz;
  tests/test_oracle_golden.py).
* Log-mel front-end: PARITY UNPINNED.  The arithmetic lives in
  torchaudio==0.4.0 (requirements.txt:12; call site tal/asr/models.py:24-32,45)
  which is neither vendored in the reference nor installed here.  It is
  restated from its published algorithm (Spectrogram = torch.stft(center=True,
  reflect, periodic Hann, onesided) -> re^2+im^2; MelScale = HTK triangular
  filterbank, f_min 0, f_max sr/2, no area normalisation) and cross-checked
  against an independent float64 numpy rfft implementation and
  transformers.audio_utils.mel_filter_bank (tests/test_oracle_logmel.py).
* generate / generate_unaligned control flow: pinned by fixtures recorded from
  the reference's own tal/asr/system.py functions driven with stubbed
  third-party imports (tests/golden/make_golden.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

SR = 16000
N_FFT = 400
HOP = 160
N_MELS = 80
N_FREQS = N_FFT // 2 + 1
KSIZE = 21


def _t(x, dtype=torch.float32):
    if isinstance(x, torch.Tensor):
        return x.to(dtype)
    return torch.from_numpy(np.ascontiguousarray(x)).to(dtype)


# ----------------------------------------------------------------------------
# Log-mel front-end (tal/asr/models.py:15-53 + torchaudio 0.4.0 semantics)
# ----------------------------------------------------------------------------
def hann_window(n=N_FFT):
    """Periodic Hann exactly as torchaudio's Spectrogram builds it: window_fn=torch.hann_window
    evaluated in float32 (differs from a float64 evaluation by up to 2.4e-7)."""
    return torch.hann_window(n)


def mel_filterbank(n_freqs=N_FREQS, n_mels=N_MELS, sr=SR, f_min=0.0, f_max=None):
    """[n_freqs, n_mels] HTK triangular filters, no area normalisation.

    Restates torchaudio 0.4.0 functional.create_fb_matrix as used by
    MelScale(n_mels=80, sample_rate=16000, f_min=0, f_max=None->sr//2)
    (tal/asr/models.py:24-32).  float32 arithmetic like the original."""
    f_max = float(sr // 2) if f_max is None else float(f_max)
    all_freqs = torch.linspace(0, sr // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)  # [n_freqs, n_mels+2]
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


def num_frames(L):
    """T = 1 + L // hop for center=True (SURVEY.md section 2 op table [probe])."""
    return 1 + int(L) // HOP


def power_spectrogram(audio):
    """[B, L] -> [B, 201, T] power spectrum (re^2 + im^2)."""
    audio = _t(audio)
    spec = torch.stft(audio, N_FFT, hop_length=HOP, win_length=N_FFT,
                      window=hann_window(), center=True, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True)
    return spec.real * spec.real + spec.imag * spec.imag


def logmel(audio, eps=1e-6, subtract_mean=True, fb=None):
    """LogMelSpec.forward (tal/asr/models.py:35-53): [B, L] -> [B, T, 80].

    mel = fb^T . power; permute to [B, T, 80]; log(mel + eps); subtract ONE
    scalar mean over the whole [B, T, 80] tensor (padding included, :52)."""
    p = power_spectrogram(audio)                      # [B, 201, T]
    fb = mel_filterbank() if fb is None else _t(fb)
    mel = torch.matmul(p.transpose(1, 2), fb)         # [B, T, 80]
    mel = torch.log(mel + eps)
    if subtract_mean:
        mel = mel - mel.mean()
    return mel


def logmel_f64_frames(audio_1d, f0, f1, eps=1e-6):
    """Frames [f0, f1) of log(mel + eps) (NO mean subtraction) of ONE clip in float64, without materialising the whole
    clip's frames: frame t covers padded samples [160 t, 160 t + 400) of the reflect-padded (200 each side) signal.
    Same arithmetic as logmel_f64; used to check hour-long clips piecewise.  -> [f1 - f0, 80] float64."""
    a = np.asarray(audio_1d, dtype=np.float64)
    L = a.shape[0]
    pad = N_FFT // 2
    lo, hi = f0 * HOP - pad, (f1 - 1) * HOP + N_FFT - pad        # sample range in unpadded coordinates, [lo, hi)
    idx = np.arange(lo, hi)
    idx = np.where(idx < 0, -idx, idx)                          # reflect (no edge repeat), as np.pad mode="reflect"
    idx = np.where(idx >= L, 2 * (L - 1) - idx, idx)
    seg = a[idx]
    k = np.arange(N_FFT, dtype=np.float64)
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * k / N_FFT)
    fi = np.arange(f1 - f0)[:, None] * HOP + np.arange(N_FFT)[None, :]
    spec = np.fft.rfft(seg[fi] * win, axis=-1)
    power = spec.real ** 2 + spec.imag ** 2
    fb = mel_filterbank().double().numpy()
    return np.log(power @ fb + eps)


def logmel_f64(audio, eps=1e-6, subtract_mean=True):
    """Independent float64 numpy restatement (explicit reflect pad + rfft)."""
    a = np.asarray(audio, dtype=np.float64)
    B, L = a.shape
    T = num_frames(L)
    pad = N_FFT // 2
    ap = np.pad(a, ((0, 0), (pad, pad)), mode="reflect")
    k = np.arange(N_FFT, dtype=np.float64)
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * k / N_FFT)
    idx = np.arange(T)[:, None] * HOP + np.arange(N_FFT)[None, :]
    frames = ap[:, idx] * win                         # [B, T, 400]
    spec = np.fft.rfft(frames, axis=-1)
    power = spec.real ** 2 + spec.imag ** 2           # [B, T, 201]
    fb = mel_filterbank().double().numpy()
    mel = np.log(power @ fb + eps)
    if subtract_mean:
        mel = mel - mel.mean()
    return mel


# ----------------------------------------------------------------------------
# TDS encoder (tal/asr/models.py:298-331 TDSBlock, :349-397 TDS)
# ----------------------------------------------------------------------------
def tds_out_len(T):
    """Length after one stride-2 k=21 pad-0 conv (tal/asr/models.py:363-364)."""
    return (int(T) - KSIZE) // 2 + 1


def tds_total_out_len(T, n_stages=3):
    for _ in range(n_stages):
        T = tds_out_len(T)
    return T


def tds_block(x, sd, p, groups):
    """TDSBlock.forward (:323-331); dropout is identity in eval. x: [B, C, T]."""
    rw = _t(sd[p + "resweight"])
    y = F.conv1d(x, _t(sd[p + "conv.0.weight"]), _t(sd[p + "conv.0.bias"]),
                 stride=1, padding=KSIZE // 2, groups=groups)
    x = x + rw * F.relu(y)
    h = F.relu(F.conv1d(x, _t(sd[p + "fc.0.weight"]), _t(sd[p + "fc.0.bias"])))
    h = F.conv1d(h, _t(sd[p + "fc.3.weight"]), _t(sd[p + "fc.3.bias"]))
    return x + rw * h


def tds_forward(x, sd, prefix="encoder.", depths=(2, 3, 6), groups=N_MELS):
    """TDS.forward = aggregate(extract(x)) (:374-397). x: [B, n_mels, T] -> [B, C3, T']."""
    x = _t(x)
    for i, depth in enumerate(depths):
        p = "%sblocks.%d." % (prefix, i)
        x = F.conv1d(x, _t(sd[p + "0.weight"]), _t(sd[p + "0.bias"]), stride=2, groups=groups)
        for j in range(depth):
            x = tds_block(x, sd, "%s1.%d." % (p, j), groups)
    return x


def padding_mask(audio_lens, t_out):
    """encoder_padding_mask (tal/asr/models.py:176-187): integer floor division
    with the data-dependent divisor audio_lens.max() // T'.  Returns bool [B, T']."""
    lens = np.asarray(audio_lens, dtype=np.int64)
    scaled = lens // (lens.max() // int(t_out))
    mask = np.zeros((len(lens), int(t_out)), dtype=bool)
    for i, l in enumerate(scaled.tolist()):
        mask[i, l:] = True
    return mask


# ----------------------------------------------------------------------------
# SDModel (tal/asr/models.py:400-485) and reconcile.get_speaker_ids (tal/baseline/reconcile.py:76-85)
# ----------------------------------------------------------------------------
def sd_encode_features(mel, sd, audio_lens=None):
    """SDModel.encode_features (:440-463). mel [B,T,80] -> dict."""
    x = tds_forward(_t(mel).permute(0, 2, 1), sd).permute(0, 2, 1).contiguous()
    mask = None if audio_lens is None else torch.from_numpy(padding_mask(audio_lens, x.size(1)))
    return {"encoder_out": x, "encoder_padding_mask": mask}


def sd_encode(audio, sd, audio_lens=None):
    """SDModel.encode (:465-471)."""
    return sd_encode_features(logmel(audio), sd, audio_lens)


def sd_decode(enc, sd):
    """SDModel.decode (:473-481): spk_logit_proj(spk_embed_proj(x))."""
    feat = F.linear(enc["encoder_out"], _t(sd["spk_embed_proj.weight"]), _t(sd["spk_embed_proj.bias"]))
    return F.linear(feat, _t(sd["spk_logit_proj.weight"]), _t(sd["spk_logit_proj.bias"]))


def get_speaker_ids(audio, sd):
    """reconcile.get_speaker_ids (tal/baseline/reconcile.py:76-85) without the
    GPU-era .cuda().half(): returns (feat_mat [T',128], ids [T'])."""
    enc = sd_encode(audio, sd, None)
    feat = F.linear(enc["encoder_out"], _t(sd["spk_embed_proj.weight"]), _t(sd["spk_embed_proj.bias"]))
    logits = sd_decode(enc, sd)
    return feat[0].numpy(), torch.argmax(logits, dim=-1)[0].numpy()


# ----------------------------------------------------------------------------
# ASRModel encoder side (tal/asr/models.py:164-201)
# ----------------------------------------------------------------------------
def asr_encode_features(mel, sd, audio_lens=None, use_speaker_head=True):
    x = tds_forward(_t(mel).permute(0, 2, 1), sd).permute(0, 2, 1)
    spk = F.linear(x, _t(sd["spk_enc_proj.weight"]), _t(sd["spk_enc_proj.bias"])) if use_speaker_head else None
    y = F.linear(x, _t(sd["decoder_proj.weight"]), _t(sd["decoder_proj.bias"]))
    mask = None if audio_lens is None else torch.from_numpy(padding_mask(audio_lens, y.size(1)))
    return {"speaker_out": spk, "encoder_out": y, "encoder_padding_mask": mask}


def asr_encode(audio, sd, audio_lens=None, use_speaker_head=True):
    return asr_encode_features(logmel(audio), sd, audio_lens, use_speaker_head)


# ----------------------------------------------------------------------------
# Decoder (tal/modules.py:41-64, tal/asr/models.py:203-289, :488-528)
# ----------------------------------------------------------------------------
def positional_encoding(max_len, d_model):
    """PositionalEncoding buffer `pe` (tal/modules.py:45-51)."""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def mha(query, key, value, sd, p, nhead, attn_mask=None, key_padding_mask=None):
    """torch.nn.MultiheadAttention forward in eval mode, written out.

    query [U,B,E], key/value [S,B,E]; packed in-proj [3E,E]; q scaled by
    head_dim^-0.5; additive float attn_mask [U,S]; bool key_padding_mask [B,S]
    -> -inf; softmax over S; returns (out [U,B,E], weights averaged over heads
    [B,U,S]) as used at tal/asr/models.py:514-519."""
    U, B, E = query.shape
    S = key.shape[0]
    hd = E // nhead
    w = _t(sd[p + "in_proj_weight"])
    b = _t(sd[p + "in_proj_bias"])
    q = F.linear(query, w[:E], b[:E]) * (float(hd) ** -0.5)
    k = F.linear(key, w[E:2 * E], b[E:2 * E])
    v = F.linear(value, w[2 * E:], b[2 * E:])
    q = q.reshape(U, B, nhead, hd).permute(1, 2, 0, 3)   # [B,H,U,hd]
    k = k.reshape(S, B, nhead, hd).permute(1, 2, 0, 3)
    v = v.reshape(S, B, nhead, hd).permute(1, 2, 0, 3)
    scores = torch.matmul(q, k.transpose(-1, -2))         # [B,H,U,S]
    if attn_mask is not None:
        scores = scores + _t(attn_mask).view(1, 1, U, S)
    if key_padding_mask is not None:
        kpm = key_padding_mask if isinstance(key_padding_mask, torch.Tensor) else torch.from_numpy(np.asarray(key_padding_mask))
        scores = scores.masked_fill(kpm.bool().view(B, 1, 1, S), float("-inf"))
    probs = torch.softmax(scores, dim=-1)
    ctx = torch.matmul(probs, v)                           # [B,H,U,hd]
    ctx = ctx.permute(2, 0, 1, 3).reshape(U, B, E)
    out = F.linear(ctx, _t(sd[p + "out_proj.weight"]), _t(sd[p + "out_proj.bias"]))
    return out, probs.mean(dim=1)


def decoder_layer(tgt, memory, sd, p, nhead, tgt_mask=None, memory_key_padding_mask=None):
    """ModRZTXDecoderLayer.forward (tal/asr/models.py:512-528); returns
    (tgt [U,B,E], src_attn_weights [B,U,S])."""
    rw = _t(sd[p + "resweight"])
    rw_src = _t(sd[p + "resweight_src"])
    a, _ = mha(tgt, tgt, tgt, sd, p + "self_attn.", nhead, attn_mask=tgt_mask)
    tgt = tgt + a * rw
    a, w = mha(tgt, memory, memory, sd, p + "multihead_attn.", nhead,
               key_padding_mask=memory_key_padding_mask)
    tgt = tgt + a * rw_src
    h = F.relu(F.linear(tgt, _t(sd[p + "linear1.weight"]), _t(sd[p + "linear1.bias"])))
    h = F.linear(h, _t(sd[p + "linear2.weight"]), _t(sd[p + "linear2.bias"]))
    return tgt + h * rw, w


def _embed_tokens(y_prev, sd):
    """embedding -> embedding_proj -> + pe[:U] (tal/asr/models.py:218-223)."""
    y = torch.as_tensor(np.asarray(y_prev), dtype=torch.long)
    e = F.embedding(y, _t(sd["embedding.weight"]))
    e = F.linear(e, _t(sd["embedding_proj.weight"]))
    pe = _t(sd["pos_dec_encoder.pe"]) if "pos_dec_encoder.pe" in sd else positional_encoding(512, e.size(-1))
    return e + pe[: e.size(1)].unsqueeze(0)


def _causal_mask(n):
    """triu(ones, 1) -> -inf (tal/asr/models.py:229-235)."""
    m = torch.triu(torch.ones(n, n), 1)
    return m.masked_fill(m == 1, float("-inf"))


def _run_layers(y_prev, memory, mask, sd, prefix, n_layers, nhead, causal_mask):
    h = _embed_tokens(y_prev, sd).permute(1, 0, 2)
    mem = _t(memory).permute(1, 0, 2)
    tgt_mask = _causal_mask(h.size(0)) if causal_mask else None
    attn = []
    for l in range(n_layers):
        h, w = decoder_layer(h, mem, sd, "%slayers.%d." % (prefix, l), nhead, tgt_mask, mask)
        attn.append(w)
    return h.permute(1, 0, 2).contiguous(), attn


def asr_decode(y_prev, enc, sd, causal_mask=True, n_layers=4, nhead=4):
    """ASRModel.decode (tal/asr/models.py:203-247) -> (logits [B,U,V], [src_attn_weights per layer])."""
    h, attn = _run_layers(y_prev, enc["encoder_out"], enc["encoder_padding_mask"], sd,
                          "decoder.", n_layers, nhead, causal_mask)
    h = F.linear(h, _t(sd["embedding_proj.weight"]).t())
    return F.linear(h, _t(sd["embedding.weight"])), attn


def asr_decode_spk(y_prev, enc, sd, causal_mask=True, n_layers=2, nhead=4):
    """ASRModel.decode_spk (tal/asr/models.py:249-289) -> [B,U,num_speakers]."""
    h, _ = _run_layers(y_prev, enc["speaker_out"], enc["encoder_padding_mask"], sd,
                       "spk_decoder.", n_layers, nhead, causal_mask)
    h = F.linear(h, _t(sd["speaker_head.0.weight"]), _t(sd["speaker_head.0.bias"]))
    return F.linear(h, _t(sd["speaker_head.1.weight"]), _t(sd["speaker_head.1.bias"]))


# ----------------------------------------------------------------------------
# UIS-RNN CoreRNN (tal/diarization/uisrnn/uisrnn.py:20-39)
# ----------------------------------------------------------------------------
def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    """One torch.nn.GRU step (gate order r, z, n):
    r = s(Wir x + bir + Whr h + bhr); z likewise; n = tanh(Win x + bin + r*(Whn h + bhn));
    h' = (1-z)*n + z*h."""
    H = h.shape[-1]
    gi = F.linear(_t(x), _t(w_ih), _t(b_ih))
    gh = F.linear(_t(h), _t(w_hh), _t(b_hh))
    r = torch.sigmoid(gi[..., :H] + gh[..., :H])
    z = torch.sigmoid(gi[..., H:2 * H] + gh[..., H:2 * H])
    n = torch.tanh(gi[..., 2 * H:] + r * gh[..., 2 * H:])
    return (1.0 - z) * n + z * _t(h)


def core_rnn(input_seq, hidden, sd, depth=1):
    """CoreRNN.forward (:33-39): input_seq [L,B,In], hidden [depth,B,H] or None
    -> (mean [L,B,obs], hidden [depth,B,H])."""
    x = _t(input_seq)
    L, B, _ = x.shape
    H = sd["gru.weight_hh_l0"].shape[1]
    h = torch.zeros(depth, B, H) if hidden is None else _t(hidden).clone()
    outs = []
    for t in range(L):
        inp = x[t]
        for l in range(depth):
            hl = gru_cell(inp, h[l], sd["gru.weight_ih_l%d" % l], sd["gru.weight_hh_l%d" % l],
                          sd["gru.bias_ih_l%d" % l], sd["gru.bias_hh_l%d" % l])
            h[l] = hl
            inp = hl
        outs.append(inp)
    out = torch.stack(outs, 0)
    m = F.relu(F.linear(out, _t(sd["linear_mean1.weight"]), _t(sd["linear_mean1.bias"])))
    m = F.linear(m, _t(sd["linear_mean2.weight"]), _t(sd["linear_mean2.bias"]))
    return m, h


# ----------------------------------------------------------------------------
# Decode control flow helper (tal/asr/util.py:5-17)
# ----------------------------------------------------------------------------
def ngram_repeat_mask(xs, n):
    xs = np.asarray(xs)
    mask = np.zeros_like(xs)
    for i, row in enumerate(xs.tolist()):
        seen = set()
        for j in range(len(row) - n):
            ng = tuple(row[j:j + n])
            if ng in seen:
                mask[i, j:j + n] = 1
            seen.add(ng)
    return mask


# ----------------------------------------------------------------------------
# Sliding-window greedy decode of a whole episode -- the CPU leg of BASELINE.json configs[4]
# ----------------------------------------------------------------------------
def generate_unaligned(audio, generated, audio_lens, sd, eos=1, chunk_size=357, max_iters=1000000, max_positions=512,
                       thresh_prct=0.5, shift_prct=0.25, stall_patience=25, rep_n=5, skip_prct=0.1,
                       n_layers=4, nhead=4, on_step=None, lm=None, lm_weight=0.0, lm_clamp=None):
    """System.generate_unaligned (tal/asr/system.py:254-524) for ONE episode on the CPU, doing the work the reference
    does per generated token: the whole live prefix through all decoder layers (no cache: the loop decodes with
    causal_mask=False, :350-351), the memory window re-projected in every layer, the LM head over every prefix position
    (:243-246), then log_softmax / argmax of the last one (:355-387).  The waveform is rounded to fp16 first (:285).
    -> (token ids [n], recorded window starts [n-1], attention rows: list of [S] arrays).
    on_step(token, row, win, hist, n_tokens, finished): called after every step with the step's raw result and the state the
    control flow left behind (tests drive the product's host-side control flow with it).
    lm / lm_weight / lm_clamp: the shallow-fusion branch (:368-384) -- lm(tokens [1, U], causal_mask=False) -> [1, U, vocab], called
    on the prefix with ids clamped to lm_clamp = len(tokenizer) - 1; its last-position log-probabilities times lm_weight are added
    on the shared part of the two vocabularies before the arg-max.
    Pinned by tests/golden/flow_unaligned*.npz, recorded from the reference's own function (tests/test_oracle_golden.py)."""
    audio = np.asarray(audio, dtype=np.float32).astype(np.float16).astype(np.float32)          # :285
    with torch.no_grad():
        enc = asr_encode(audio, sd, audio_lens)                                                # :290
        mem_all, mask_all = enc["encoder_out"], enc["encoder_padding_mask"]
        enc_len = int((~mask_all).sum(dim=-1)[0])                                              # :291
        toks = [int(t) for t in np.asarray(generated).reshape(-1)]
        starts, rows = [], []
        win = hist = 0                      # chunk_start, history_start (:300-303)
        best = 0.0                          # highest_progress
        stale = since = 0                   # num_no_improve, window_time
        for _ in range(max_iters):
            prefix = toks[hist:]                                                               # :338
            assert len(prefix) <= max_positions, "Cannot exceed max context length"
            sl = slice(win, win + chunk_size)                                                  # python slice semantics, :347-348
            memory = {"encoder_out": mem_all[:, sl], "encoder_padding_mask": mask_all[:, sl]}
            logits, attn = asr_decode(np.asarray([prefix]), memory, sd, causal_mask=False, n_layers=n_layers, nhead=nhead)
            last = logits[:, -1, :]
            if torch.isnan(last).any():
                raise Exception("Logits contain nans!")
            logprobs = F.log_softmax(last, dim=-1)                                             # :366
            if lm is not None and lm_weight > 0:                                               # :368-384
                lm_in = torch.clamp(torch.as_tensor([prefix], dtype=torch.long), max=lm_clamp)
                lm_lp = F.log_softmax(lm(lm_in, causal_mask=False)[:, -1, :].float(), dim=-1)
                k = min(lm_lp.size(-1), logprobs.size(-1))
                logprobs[:, :k] += lm_lp[:, :k] * lm_weight
            picked = int(logprobs.argmax(dim=-1)[0])                                           # :386-387
            toks.append(picked)
            row = torch.stack(attn, dim=0).mean(dim=0)[0, -1]                                  # :392-397
            starts.append(win)
            rows.append(row.numpy().copy())
            ramp = torch.arange(row.numel()).to(row) / row.numel()                             # :405-408
            progress = float((row * ramp).sum())
            if progress > best:                                                                # :411-419
                stale = 0
                if since > 5:
                    best = progress
            else:
                stale += 1
            stalling = stale >= stall_patience
            repeating = int(ngram_repeat_mask(np.asarray([prefix]), rep_n).sum()) > rep_n * 2  # :426-429
            last_chunk = enc_len - win <= chunk_size
            reset = stalling or repeating
            kept = True
            if not last_chunk:
                if reset:                                                                      # :437-456
                    win += int(chunk_size * skip_prct)
                    if repeating:
                        back = 2 * rep_n - 1
                        del toks[-back:], starts[-back:], rows[-back:]
                        kept = False
                    toks[-1] = eos
                    hist = len(toks) - 1
                    best, since = 0.0, 0
                elif progress > thresh_prct:                                                   # :462-476
                    size = len(toks) - hist
                    win += int(chunk_size * shift_prct)
                    hist += int(torch.tensor(shift_prct / thresh_prct * (size - 1)).floor().long())
                    best, since = 0.0, 0
            if kept:
                starts[-1] = win            # the reference appends the chunk_start TENSOR and advances it in place (:400,441,468)
            win = min(win, enc_len - chunk_size)                                               # :480
            hist = max(hist, max(len(toks) - max_positions, 0))                                # :482-483
            assert hist < len(toks) and len(toks) - hist <= max_positions
            since += 1
            if on_step is not None:
                on_step(picked, row.numpy(), win, hist, len(toks), reset and last_chunk)
            if reset and last_chunk:                                                           # :510-519
                break
    return np.asarray(toks, dtype=np.int64), np.asarray(starts, dtype=np.int64), rows


# ----------------------------------------------------------------------------
# End-to-end CPU path used as bench.py's cpu_baseline ("port")
# ----------------------------------------------------------------------------
def sd_path(audio, sd):
    """waveform -> log-mel -> TDS -> SD head; returns (feat [B,T',128], logits
    [B,T',6008], ids [B,T']) -- the metric's end-to-end path."""
    with torch.no_grad():
        enc = sd_encode(audio, sd, None)
        feat = F.linear(enc["encoder_out"], _t(sd["spk_embed_proj.weight"]), _t(sd["spk_embed_proj.bias"]))
        logits = F.linear(feat, _t(sd["spk_logit_proj.weight"]), _t(sd["spk_logit_proj.bias"]))
        return feat, logits, logits.argmax(-1)
