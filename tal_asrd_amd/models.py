"""Host-side mirror of the reference's model classes (tal/asr/models.py), backed by
the HIP kernels of libtal_asrd_hip.so.

Same class names, constructor signatures, method names, return dicts and
state_dict keys as the reference, so `System.generate*`-style control flow and
checkpoint loading work unchanged:

    LogMelSpec, TDSBlock, TDS, SDModel, ASRModel, ModRZTXDecoderLayer

Differences that are deliberate and documented in DESIGN.md:
  * inference only: dropout / SpecAugment (models.py:531-566) are not built;
  * activations are time-major [B, T, C] inside; `TDS.forward` keeps the
    reference's [B, C, T] contract by transposing at its edge, while
    `encode_features` (which receives and returns time-major tensors in the
    reference too) never transposes;
  * fp32 results end to end; the waveform itself may arrive as fp16 (the reference's callers cast it: system.py:92,285,
    reconcile.py:78) -- the front-end widens it exactly and computes in fp32 / fp64 from there;
  * CPU tensors are rejected: there is no fallback path.
The nn.Conv1d / nn.Linear / nn.MultiheadAttention children are parameter
containers that keep the reference's key names and default initialisation; the
wrappers below override `forward` so that a direct call (e.g.
`model.spk_embed_proj(x)` in tal/baseline/reconcile.py:81) also runs the HIP path.
"""
import math

import torch
import torch.nn as nn

from . import _native as N
from . import ops
from . import tiling
from .modules import PositionalEncoding, weight_init

DEFAULT_SR = 16000  # tal/asr/data/__init__.py:6
KERNEL_SIZE = 21


# ----------------------------------------------------------------------------
# torchaudio-0.4.0-compatible buffers (restated; the library is not available)
# ----------------------------------------------------------------------------
def _htk_filterbank(n_freqs, n_mels, sample_rate, f_min=0.0, f_max=None):
    """[n_freqs, n_mels] triangular HTK-mel filters without area normalisation
    (torchaudio 0.4.0 MelScale / create_fb_matrix semantics)."""
    f_max = float(sample_rate // 2) if f_max is None else float(f_max)
    freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    lo = 2595.0 * math.log10(1.0 + f_min / 700.0)
    hi = 2595.0 * math.log10(1.0 + f_max / 700.0)
    edges_mel = torch.linspace(lo, hi, n_mels + 2)
    edges_hz = 700.0 * (10.0 ** (edges_mel / 2595.0) - 1.0)
    width = edges_hz[1:] - edges_hz[:-1]
    dist = edges_hz.unsqueeze(0) - freqs.unsqueeze(1)
    falling = (-1.0 * dist[:, :-2]) / width[:-1]
    rising = dist[:, 2:] / width[1:]
    return torch.clamp(torch.min(falling, rising), min=0.0)


# Cached parameter lists (TDS._param_key, decoder.layer_weights) must notice a REPLACED Parameter object (`layer.weight =
# nn.Parameter(...)`, load_state_dict(assign=True): both go through Module.register_parameter).  The counter that tells them is
# per module TREE of this package: `_adopt(root)` puts one shared cell into the __dict__ of every module of the tree, and the
# registration hook -- installed when the first tree is adopted, not at import -- bumps the cell of the module it is called for
# and touches nothing else: modules of the host program (the reference's training code, Lightning) carry no cell.
_EPOCH_ATTR = "_tal_param_epoch"
_hook_installed = [False]


def _note_parameter_registration(module, name, param):
    cell = module.__dict__.get(_EPOCH_ATTR)
    if cell is not None:
        cell[0] += 1


def _note_module_registration(module, name, child):
    # a submodule attached (or replaced) after the tree was adopted: it joins the tree's counter, and the counter moves -- the
    # cached parameter lists of the tree were built without it
    cell = module.__dict__.get(_EPOCH_ATTR)
    if cell is not None and child is not None:
        for m in child.modules():
            m.__dict__[_EPOCH_ATTR] = cell
        cell[0] += 1


def _install_hooks():
    """Idempotent; called wherever a counter is read or handed out -- a module tree restored by torch.load / pickle in a fresh
    process carries its cell in the modules' __dict__ without any constructor of this package having run."""
    if not _hook_installed[0]:
        torch.nn.modules.module.register_module_parameter_registration_hook(_note_parameter_registration)
        torch.nn.modules.module.register_module_module_registration_hook(_note_module_registration)
        _hook_installed[0] = True


def _adopt(root):
    """Give every module of `root`'s tree the tree's parameter-registration counter; -> the counter cell."""
    _install_hooks()
    cell = root.__dict__.get(_EPOCH_ATTR)
    if cell is None:
        cell = [0]
    for m in root.modules():
        m.__dict__[_EPOCH_ATTR] = cell
    return cell


def param_epoch(root):
    """The tree's counter value (the tree is adopted on first use: modules built by other code, e.g. a deep copy, too)."""
    _install_hooks()
    cell = root.__dict__.get(_EPOCH_ATTR)
    if cell is None:
        cell = _adopt(root)
    return cell[0]


class _Spectrogram(nn.Module):
    def __init__(self, n_fft):
        super().__init__()
        self.register_buffer("window", torch.hann_window(n_fft))


class _MelScale(nn.Module):
    def __init__(self, n_mels, sample_rate, n_freqs):
        super().__init__()
        self.register_buffer("fb", _htk_filterbank(n_freqs, n_mels, sample_rate))


class MelSpectrogram(nn.Module):
    """Buffer holder with torchaudio's attribute paths (`spectrogram.window`,
    `mel_scale.fb`), so reference checkpoints load with strict=True."""

    def __init__(self, sample_rate, n_mels, n_fft, win_length, hop_length):
        super().__init__()
        if (sample_rate, n_fft, win_length, hop_length, n_mels) != (16000, 400, 400, 160, 80):
            raise N.NativeError("the HIP front-end is built for 16 kHz / 400 / 400 / 160 / 80 mels "
                                "(tal/asr/models.py:24-32)")
        self.spectrogram = _Spectrogram(n_fft)
        self.mel_scale = _MelScale(n_mels, sample_rate, n_fft // 2 + 1)


class LogMelSpec(nn.Module):
    """tal/asr/models.py:15-53."""

    def __init__(self, sr=DEFAULT_SR, n_mels=80, eps=1e-6):
        super().__init__()
        self.mel_transform = MelSpectrogram(sample_rate=sr, n_mels=n_mels, n_fft=int(25 / 1000 * sr),
                                            win_length=int(25 / 1000 * sr), hop_length=int(10 / 1000 * sr))
        self.eps = eps
        self._plan = None
        self._plan_key = None

    def plan(self):
        win, fb = self.mel_transform.spectrogram.window, self.mel_transform.mel_scale.fb
        key = (win.data_ptr(), win._version, fb.data_ptr(), fb._version)
        if self._plan is None or self._plan_key != key:
            self._plan = ops.logmel_plan(win, fb)
            self._plan_key = key
        return self._plan

    @torch.no_grad()
    def forward(self, audio: torch.Tensor):
        """audio [batch, audio_len] (fp32 or fp16) -> [batch, frames, n_mels] fp32, minus the global mean of this call."""
        N.require_cuda(audio, "LogMelSpec.forward")
        return ops.logmel(self.plan(), audio, eps=self.eps, subtract_mean=True)

    def forward_unsubtracted(self, audio: torch.Tensor):
        """-> (log-mel [batch, frames, n_mels] BEFORE the mean subtraction, its global mean as a device tensor [1]): for callers that
        hand both to TDS.forward_then(x_mean=...), where the subtraction becomes a bias correction of the first resize conv."""
        N.require_cuda(audio, "LogMelSpec.forward_unsubtracted")
        out, mean, _ = ops.logmel(self.plan(), audio, eps=self.eps, subtract_mean=False, return_stats=True)
        return out, mean


# ----------------------------------------------------------------------------
# parameter containers whose direct call runs the HIP path
# ----------------------------------------------------------------------------
class Linear(nn.Linear):
    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


class PointwiseConv1d(nn.Conv1d):
    """Conv1d(k=1) == dense layer over channels; called on time-major [B, T, C]."""

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


class GroupedConv1d(nn.Conv1d):
    """Conv1d(k=21, groups=G) holder; `packed()` caches the kernel-side weight layout."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._packed = None
        self._packed_key = None

    def packed(self):
        key = (self.weight.data_ptr(), self.weight._version, self.weight.device)
        if self._packed is None or self._packed_key != key:
            self._packed = ops.pack_gconv_weight(self.weight.detach(), self.groups)
            self._packed_key = key
        return self._packed

    def forward(self, x):
        """time-major [B, T, C_in] -> [B, T_out, C_out] (plain conv + bias, no activation)."""
        if self.stride[0] != 2 or self.padding[0] != 0:
            raise N.NativeError("GroupedConv1d.forward: only the stride-2 / pad-0 resize conv is exposed directly")
        return ops.gconv_s2(x, self.packed(), self.bias, self.out_channels, self.groups)


class TDSBlock(nn.Module):
    """tal/asr/models.py:298-331: x + rw*relu(gconv(x)); x + rw*fc(x)."""

    def __init__(self, hidden, kernel_size, groups, dropout=0.1):
        super().__init__()
        if kernel_size != KERNEL_SIZE:
            raise N.NativeError("TDSBlock: the HIP kernels are built for kernel_size=21")
        self.conv = nn.Sequential(
            GroupedConv1d(hidden, hidden, kernel_size=kernel_size, stride=1, groups=groups,
                          padding=kernel_size // 2),
            nn.ReLU(), nn.Dropout(dropout))
        self.fc = nn.Sequential(
            PointwiseConv1d(hidden, hidden, kernel_size=1, stride=1), nn.ReLU(), nn.Dropout(dropout),
            PointwiseConv1d(hidden, hidden, kernel_size=1, stride=1), nn.Dropout(dropout))
        self.resweight = nn.Parameter(torch.Tensor([0]))

    def forward_time_major(self, x):
        rw = float(self.resweight.detach())
        g = self.conv[0]
        x = ops.gconv_res(x, g.packed(), g.bias, rw, g.groups)
        h = ops.linear(x, self.fc[0].weight, self.fc[0].bias, mode=1)
        return ops.linear(h, self.fc[3].weight, self.fc[3].bias, mode=2, res=x, alpha=rw)

    def forward(self, x):
        """x [batch, features, time] as in the reference."""
        return self.forward_time_major(x.permute(0, 2, 1).contiguous()).permute(0, 2, 1).contiguous()


class TDS(nn.Module):
    """tal/asr/models.py:349-397.  `forward` keeps the [B, C, T] contract;
    `forward_time_major` is the native entry ([B, T, C] in and out, one C call)."""

    def __init__(self, input_size, sizes, depths, kernel_size=21, dropout=0.1):
        super().__init__()
        if kernel_size != KERNEL_SIZE:
            raise N.NativeError("TDS: the HIP kernels are built for kernel_size=21")
        if len(sizes) - 1 > N.TAL_MAX_STAGES or max(depths) > N.TAL_MAX_DEPTH:
            raise N.NativeError("TDS: at most %d stages of depth %d" % (N.TAL_MAX_STAGES, N.TAL_MAX_DEPTH))
        self.extract_block_id = 1
        self.sizes = sizes
        self.depths = list(depths)
        self.max_item_frames = tiling.max_item_frames(list(sizes))     # longer single items are encoded tile by tile
        self.tile_frames = 32768                            # output frames per tile then (4.4 min of audio)
        self.input_size = input_size
        self.blocks = nn.Sequential(*[
            nn.Sequential(
                GroupedConv1d(sizes[i - 1], sizes[i], kernel_size=kernel_size, stride=2, groups=input_size),
                nn.Sequential(*[TDSBlock(sizes[i], kernel_size, input_size, dropout=dropout)
                                for _ in range(depths[i - 1])]))
            for i in range(1, len(sizes))])
        self._plist = None
        self._plist_epoch = -1
        _adopt(self)
        self._descs = {}        # (first, last) -> tal_tds_desc
        self._packs = {}        # stage -> packed / split weights (kept alive here)
        self._desc_key = None   # parameter versions the caches were built for

    def _apply(self, fn, *a, **kw):
        self._plist = None          # .to() / .cuda() may replace the Parameter objects
        return super()._apply(fn, *a, **kw)


    def _param_key(self):
        # (walking the module tree costs ~200 us per call, the cached list ~15 us: it matters for 30-second clips)
        # The list is rebuilt whenever a module of THIS tree registered a Parameter since it was taken (param_epoch: direct
        # `layer.weight = nn.Parameter(...)` assignments and load_state_dict(assign=True) go through register_parameter)
        # or this module was converted (_apply); in-place updates show in p._version, storage moves in data_ptr().
        epoch = param_epoch(self)
        if self._plist is None or self._plist_epoch != epoch:
            self._plist = list(self.parameters())
            self._plist_epoch = epoch
        return tuple((p.data_ptr(), p._version) for p in self._plist)

    def _stage_pack(self, s):
        """Kernel-side weight forms of stage s (resize conv + its TDSBlocks), built once per parameter version and
        shared by every descriptor that covers the stage (forward / extract / aggregate)."""
        if s in self._packs:
            return self._packs[s]
        down, chain = self.blocks[s][0], self.blocks[s][1]
        c = self.sizes[s + 1]
        # fp16-range guard, weight side: a weight beyond the finite fp16 range (or non-finite) cannot be carried as hi / lo
        # halves; such a layer keeps the exact fp32 kernels (no split / fragment form is built for it)
        def in_range(w):
            return bool(torch.isfinite(w).all()) and float(w.abs().max()) <= 65504.0
        pack = {"down_w": down.packed(),
                "down_frag": ops.pack_gconv_f16x3_weight(down.weight.detach(), down.groups, stride=2) if in_range(down.weight.detach()) else None,
                "blocks": []}
        rws = torch.stack([blk.resweight.detach().reshape(()) for blk in chain]).cpu().tolist() if len(chain) else []
        for blk, rw in zip(chain, rws):
            g = blk.conv[0]
            b = {"conv_w": g.packed(), "rw": float(rw), "fc0_split": None, "fc3_split": None, "conv_frag": None}
            if c % 160 == 0 and in_range(blk.fc[0].weight.detach()) and in_range(blk.fc[3].weight.detach()):
                # hi / lo fp16 splits of the two pointwise weights: long inputs run these layers in the fp16x3
                # form (include/tal_asrd.h), fp32-equivalent results at ~2.4x the fp32 matrix rate
                b["fc0_split"] = ops.split_f16x3(blk.fc[0].weight.detach().reshape(c, c))
                b["fc3_split"] = ops.split_f16x3(blk.fc[3].weight.detach().reshape(c, c))
            if in_range(g.weight.detach()):
                # ... and the grouped conv as fp16x3 MFMA operand fragments (widths 10 / 14 / 18 per group)
                b["conv_frag"] = ops.pack_gconv_f16x3_weight(g.weight.detach(), g.groups)
            pack["blocks"].append(b)
        self._packs[s] = pack
        return pack

    def _descriptor(self, first=0, last=None):
        """Build (and cache per (first, last)) the tal_tds_desc for stages [first, last)."""
        last = len(self.sizes) - 1 if last is None else last
        key = self._param_key()
        if key != self._desc_key:          # a parameter changed (load_state_dict, .to(device)): drop every cached form
            self._descs, self._packs, self._desc_key = {}, {}, key
        if (first, last) in self._descs:
            return self._descs[(first, last)]
        d = N.TdsDesc()
        d.n_stages = last - first
        d.groups = self.input_size
        for s in range(first, last + 1):
            d.channels[s - first] = self.sizes[s]
        for s in range(first, last):
            i = s - first
            down, chain = self.blocks[s][0], self.blocks[s][1]
            pack = self._stage_pack(s)
            d.depths[i] = len(chain)
            d.down_w[i] = pack["down_w"].data_ptr()
            d.down_b[i] = down.bias.data_ptr()
            if pack["down_frag"] is not None:
                d.down_w_frag[i] = pack["down_frag"].data_ptr()
            for j, (blk, b) in enumerate(zip(chain, pack["blocks"])):
                g = blk.conv[0]
                bw = d.blocks[i][j]
                bw.conv_w, bw.conv_b = b["conv_w"].data_ptr(), g.bias.data_ptr()
                bw.fc0_w, bw.fc0_b = blk.fc[0].weight.data_ptr(), blk.fc[0].bias.data_ptr()
                bw.fc3_w, bw.fc3_b = blk.fc[3].weight.data_ptr(), blk.fc[3].bias.data_ptr()
                bw.resweight = b["rw"]
                if b["fc0_split"] is not None:
                    bw.fc0_w_split, bw.fc3_w_split = b["fc0_split"].data_ptr(), b["fc3_split"].data_ptr()
                if b["conv_frag"] is not None:
                    bw.conv_w_frag = b["conv_frag"].data_ptr()
        self._descs[(first, last)] = d
        return d

    def _needs_tiles(self, x, first, last):
        """One item longer than the kernels' 2 GiB-per-item limit (~3.7 h of audio): the whole stack runs tile by tile with
        the receptive-field halo (tiling.py) instead of falling back to the slow generic kernels."""
        return first == 0 and last == len(self.sizes) - 1 and x.shape[0] == 1 and x.shape[1] > self.max_item_frames

    def forward_time_major(self, x, first=0, last=None):
        N.require_cuda(x, "TDS.forward")
        last = len(self.sizes) - 1 if last is None else last
        if first == last:
            return x
        if self._needs_tiles(x, first, last):
            # one item beyond the kernels' per-item limit: tile by tile inside the C call (tal_tds_tiled_fwd)
            return ops.tds_forward_tiled(self._descriptor(first, last), x, self.sizes[last], self.tile_frames)
        return ops.tds_forward(self._descriptor(first, last), x, self.sizes[last])

    def forward_then(self, x, tail, x_mean=None, split_ok=False):
        """tail(encoder output) with the fp16-range check of the encoder call read AFTER tail's kernels are enqueued (the
        read waits for the stream: done first, it would leave the GPU idle while the host launches the heads).  If the
        check fires, the encoder is re-run on the exact fp32 kernels and tail is applied again.
        x_mean: x is LogMelSpec.forward_unsubtracted's tensor and x_mean its mean (folded into the first resize conv where the
        kernels can -- ops.tds_premean_ok --, subtracted here otherwise).
        split_ok: `tail` is called as tail(y, y_split) and can take the encoder output in the hi / lo split form (y_split True:
        long inputs, where the last stage runs all-split); after an exact re-run it is called with the fp32 output."""
        N.require_cuda(x, "TDS.forward")
        desc = self._descriptor(0, len(self.sizes) - 1)
        if x_mean is not None and (self._needs_tiles(x, 0, len(self.sizes) - 1) or not ops.tds_premean_ok(desc, x)):
            x, x_mean = ops.subtract_scalar_(x, x_mean), None
        if self._needs_tiles(x, 0, len(self.sizes) - 1):
            y = self.forward_time_major(x)
            return tail(y, False) if split_ok else tail(y)
        y, chk = ops.tds_forward(desc, x, self.sizes[-1], defer=True, x_mean=x_mean, out_split=split_ok)
        out = tail(y, chk.y_split) if split_ok else tail(y)
        if chk.flagged():
            y = chk.rerun_exact()
            out = tail(y, False) if split_ok else tail(y)
        return out

    def extract(self, x):
        """Feature extraction network: blocks[:extract_block_id] ([B, C, T])."""
        y = self.forward_time_major(x.permute(0, 2, 1).contiguous(), 0, self.extract_block_id)
        return y.permute(0, 2, 1).contiguous()

    def aggregate(self, x):
        y = self.forward_time_major(x.permute(0, 2, 1).contiguous(), self.extract_block_id, len(self.sizes) - 1)
        return y.permute(0, 2, 1).contiguous()

    def forward(self, x):
        """x: mel features [batch, features, time] -> [batch, sizes[-1], time']."""
        return self.forward_time_major(x.permute(0, 2, 1).contiguous()).permute(0, 2, 1).contiguous()


def padding_mask(audio_lens, t_out, device):
    """tal/asr/models.py:176-187, literally: scaled = lens // (lens.max() // T'); mask[i, scaled_i:] = 1.
    Integer math on the host (the reference loops over .tolist() as well)."""
    lens = audio_lens.detach().cpu().to(torch.int64)
    scaled = lens // (lens.max() // int(t_out))
    mask = torch.zeros(lens.numel(), int(t_out), dtype=torch.bool)
    for i, l in enumerate(scaled.tolist()):
        mask[i, l:] = 1
    return mask.to(device)


class SDModel(nn.Module):
    """Separate-diarizer baseline (tal/asr/models.py:400-485)."""

    def __init__(self, num_speakers=6008, n_mels=80, dropout=0.2, embed_size=128):
        super().__init__()
        self.num_speakers = num_speakers
        tds_sizes = [n_mels, 10 * n_mels, 14 * n_mels, 18 * n_mels]
        tds_depths = [2, 3, 6]
        self.logmelspec = LogMelSpec(n_mels=n_mels)
        self.encoder = TDS(n_mels, tds_sizes, tds_depths, dropout=dropout)
        self.dropout = nn.Dropout(dropout)
        self.spk_embed_proj = Linear(tds_sizes[-1], embed_size)
        self.spk_logit_proj = Linear(embed_size, num_speakers)
        self.apply(weight_init())
        self.eval()

    def get_encoder_params(self):
        return list(self.encoder.parameters())

    def extract_features(self, x, specaug=True):
        return self.logmelspec(x)

    def encode_features(self, x: torch.Tensor, audio_lens: torch.LongTensor = None):
        x = self.encoder.forward_time_major(x)
        mask = None if audio_lens is None else padding_mask(audio_lens, x.size(1), x.device)
        return {"encoder_out": x, "encoder_padding_mask": mask}

    def encode(self, x: torch.Tensor, audio_lens: torch.LongTensor = None):
        return self.encode_features(self.extract_features(x), audio_lens)

    def decode(self, encoder_out, past=None, causal_mask=True):
        _, logits, _ = ops.sd_head(encoder_out["encoder_out"], self.spk_embed_proj.weight, self.spk_embed_proj.bias,
                                   self.spk_logit_proj.weight, self.spk_logit_proj.bias,
                                   want_logits=True, want_ids=False)
        return logits

    def forward(self, x, audio_lens):
        encoder_out = self.encode(x, audio_lens)
        return self.decode(encoder_out), encoder_out

    @torch.no_grad()
    def speaker_ids(self, x_wav, want_logits=False):
        """The fused form of tal/baseline/reconcile.py:76-85 (get_speaker_ids): whole-episode
        waveform [1, L] -> (feat [T', 128], ids [T'] int32[, logits]) without materialising
        the [T', 6008] logits unless asked."""
        mel, mean = self.logmelspec.forward_unsubtracted(x_wav)
        return self.speaker_ids_from_logmel(mel, mean, want_logits=want_logits)

    @torch.no_grad()
    def speaker_ids_from_logmel(self, mel, mean, want_logits=False):
        """The same from the log-mel BEFORE its mean subtraction and the scalar to subtract (a device tensor [1]): for callers
        that own the mean -- a reference call whose batch is spread over several GPUs subtracts the mean of the WHOLE call
        (tal/asr/models.py:52), i.e. an all-reduced (sum, count), distributed.allreduce_logmel_stats.  The subtraction rides in the
        first resize conv's bias (eval: extract_features is the log-mel alone, models.py:430-438)."""
        def head(enc_out, enc_split=False):
            # (long inputs: the encoder output arrives in the hi / lo split form and the 1440 -> 128 embedding layer runs in the
            #  fp16x3 form on it, tal_sd_head_split_fwd; its weight split is cached per parameter version)
            return ops.sd_head(enc_out, self.spk_embed_proj.weight, self.spk_embed_proj.bias, self.spk_logit_proj.weight,
                               self.spk_logit_proj.bias, want_logits=want_logits, want_ids=True, x_split=enc_split,
                               w_embed_split=self._embed_split() if enc_split else None)
        feat, logits, ids = self.encoder.forward_then(mel, head, x_mean=mean, split_ok=self._embed_split() is not None)
        return (feat, ids, logits) if want_logits else (feat, ids)

    def _embed_split(self):
        """hi / lo fp16 split of spk_embed_proj.weight (None: a weight outside the fp16 range, or a width the fp16x3 layer does not
        take -- the head then runs its fp32 embedding layer on the fp32 encoder output)."""
        w = self.spk_embed_proj.weight
        key = (w.data_ptr(), w._version, w.device)
        if getattr(self, "_embed_split_key", None) != key:
            wd = w.detach()
            ok = w.is_cuda and w.shape[1] % 32 == 0 and bool(torch.isfinite(wd).all()) and float(wd.abs().max()) <= 65504.0
            self._embed_split_t = ops.split_f16x3(wd.contiguous()) if ok else None
            self._embed_split_key = key
        return self._embed_split_t

    @torch.no_grad()
    def speaker_ids_stream(self, host_clips):
        """The loop of tal/baseline/reconcile.py:96-102 (one episode after the other: load, `.cuda()`, get_speaker_ids) over
        waveforms held in host memory, with the upload of episode i + 1 on a copy stream under the compute of episode i:
        `host_clips` = iterable of [1, L] float32 (or float16) tensors (pinned memory for a truly asynchronous copy); yields
        (feat, ids) per clip, in order.  With ~230 MB per hour of audio and ~50 GB/s of PCIe the copy (4.5 ms) hides
        entirely behind the 18 ms of compute."""
        dev = self.spk_embed_proj.weight.device
        copy_stream = torch.cuda.Stream(device=dev)
        compute = torch.cuda.current_stream(dev)
        # exactly two device buffers (slot 0 / 1), flat and grow-only: sized to the longest clip seen so far, a clip lands in a
        # view of its slot.  A fresh allocation per clip would be a hipMalloc (a device synchronisation) in the middle of the
        # stream; one cached buffer per clip SHAPE grows without bound over a corpus whose episodes all differ in length.
        if not hasattr(self, "_stream_bufs"):
            self._stream_bufs = [None, None]    # kept between calls
        bufs, free_ev = self._stream_bufs, [None, None]

        def upload(clip, slot):
            if clip.dtype not in (torch.float32, torch.float16):
                raise N.NativeError("speaker_ids_stream: clips must be float32 or float16 waveforms, got %s" % clip.dtype)
            nbytes = clip.numel() * clip.element_size()
            if bufs[slot] is None or bufs[slot].numel() < nbytes:
                if free_ev[slot] is not None:
                    free_ev[slot].synchronize()               # (the old buffer's last reader; rare: only when a longer clip arrives)
                bufs[slot] = torch.empty(nbytes + nbytes // 8, dtype=torch.uint8, device=dev)
                free_ev[slot] = None
            view = bufs[slot][:nbytes].view(clip.dtype).view(clip.shape)
            if free_ev[slot] is not None:
                copy_stream.wait_event(free_ev[slot])         # the compute that last read this buffer is done
            else:
                copy_stream.wait_stream(compute)              # (first use: the allocation above is ordered on the compute stream)
            with torch.cuda.stream(copy_stream):
                view.copy_(clip, non_blocking=True)
                done = torch.cuda.Event()
                done.record(copy_stream)
            return view, done, slot

        it = iter(host_clips)
        first = next(it, None)
        if first is None:
            return
        pending = upload(first, 0)
        i = 0
        while pending is not None:
            x, done, slot = pending
            nxt = next(it, None)
            pending = upload(nxt, (i + 1) & 1) if nxt is not None else None
            compute.wait_event(done)
            out = self.speaker_ids(x)
            ev = torch.cuda.Event()
            ev.record(compute)
            free_ev[slot] = ev
            i += 1
            yield out


class ModRZTXDecoderLayer(nn.Module):
    """ReZero decoder layer that caches cross-attention weights (tal/asr/models.py:488-528).
    Parameter layout identical to the reference (nn.MultiheadAttention packed in-proj)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu"):
        super().__init__()
        if activation != "relu":
            raise N.NativeError("ModRZTXDecoderLayer: only the ReLU FFN is built (tal/asr/models.py:125)")
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = Linear(dim_feedforward, d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.dropout3 = nn.Dropout(dropout)
        self.resweight = nn.Parameter(torch.Tensor([0]))
        self.resweight_src = nn.Parameter(torch.Tensor([0]))
        self.nhead = nhead
        self.src_attn_weights = None
        _adopt(self)

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None):
        from .decoder import decoder_layer_forward
        return decoder_layer_forward(self, tgt, memory, tgt_mask, memory_mask, tgt_key_padding_mask,
                                     memory_key_padding_mask)


class ASRModel(nn.Module):
    """Joint ASR + speaker model (tal/asr/models.py:56-295)."""

    def __init__(self, model_type="2x", num_speakers=0, n_mels=80, vocab_size=10000, n_head=4, max_positions=512,
                 dropout=0.2, embed_size=64, spk_embed=128, use_speaker_head=False):
        super().__init__()
        self.embed_size = embed_size
        self.num_speakers = num_speakers
        self.max_positions = max_positions
        self.use_speaker_head = use_speaker_head
        self.model_type = model_type
        self.n_head = n_head
        tds_sizes = [n_mels, 10 * n_mels, 14 * n_mels, 18 * n_mels]
        tds_depths = [2, 3, 6]
        if model_type == "1x":
            d_hidden, n_layers = 256, 4
        elif model_type == "2x":
            d_hidden, n_layers = 512, 4
        else:
            raise Exception("Invalid model type")
        self.pos_dec_encoder = PositionalEncoding(d_hidden, max_len=max_positions, dropout=dropout)
        self.logmelspec = LogMelSpec(n_mels=n_mels)
        self.encoder = TDS(n_mels, tds_sizes, tds_depths, dropout=dropout)
        self.dropout = nn.Dropout(dropout)
        self.decoder_proj = Linear(tds_sizes[-1], d_hidden)
        num_tokens = vocab_size if use_speaker_head else vocab_size + num_speakers
        if self.embed_size:
            self.embedding = nn.Embedding(num_tokens, embed_size)
            self.embedding_proj = Linear(embed_size, d_hidden, bias=False)
            self.lm_head = Linear(embed_size, num_tokens, bias=False)
        else:
            self.embedding = nn.Embedding(num_tokens, d_hidden)
            self.lm_head = Linear(d_hidden, num_tokens, bias=False)
        self.lm_head.weight = self.embedding.weight  # tied (models.py:117)
        self.decoder = _decoder_stack(d_hidden, n_head, dropout, n_layers)
        if self.use_speaker_head:
            self.spk_enc_proj = Linear(tds_sizes[-1], d_hidden)
            self.spk_decoder = _decoder_stack(d_hidden, n_head, dropout, n_layers // 2)
            self.speaker_head = nn.Sequential(Linear(d_hidden, spk_embed), Linear(spk_embed, num_speakers))
        self.apply(weight_init())
        self.eval()

    def get_encoder_params(self):
        return list(self.encoder.parameters())

    def extract_features(self, x, specaug=True):
        return self.logmelspec(x)

    def encode_features(self, x: torch.Tensor, audio_lens: torch.LongTensor = None):
        def proj(e):
            spk = ops.linear(e, self.spk_enc_proj.weight, self.spk_enc_proj.bias) if self.use_speaker_head else None
            return spk, ops.linear(e, self.decoder_proj.weight, self.decoder_proj.bias)
        spk_h, x = self.encoder.forward_then(x, proj)
        mask = None if audio_lens is None else padding_mask(audio_lens, x.size(1), x.device)
        return {"speaker_out": spk_h, "encoder_out": x, "encoder_padding_mask": mask}

    def encode(self, x: torch.Tensor, audio_lens: torch.LongTensor = None):
        return self.encode_features(self.extract_features(x), audio_lens)

    def decode(self, y_prev, encoder_out, past=None, causal_mask=True):
        from .decoder import asr_decode
        return asr_decode(self, y_prev, encoder_out, causal_mask)

    def decode_spk(self, y_prev, encoder_out, causal_mask=True):
        from .decoder import asr_decode_spk
        return asr_decode_spk(self, y_prev, encoder_out, causal_mask)

    def forward(self, x, y_prev, audio_lens):
        encoder_out = self.encode(x, audio_lens)
        lm_out = self.decode(y_prev, encoder_out)
        spk_out = self.decode_spk(y_prev, encoder_out) if self.use_speaker_head else None
        return (lm_out, spk_out), encoder_out


class TransformerDecoder(nn.Module):
    """Layer stack with the key layout of nn.TransformerDecoder (`layers.N.*`, norm=None) and the
    plain torch-1.4 loop the reference was written against."""

    def __init__(self, layers):
        super().__init__()
        self.layers = nn.ModuleList(layers)
        self.num_layers = len(layers)
        self.norm = None

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None):
        out = tgt
        for layer in self.layers:
            out = layer(out, memory, tgt_mask=tgt_mask, memory_mask=memory_mask,
                        tgt_key_padding_mask=tgt_key_padding_mask,
                        memory_key_padding_mask=memory_key_padding_mask)
        return out


def _decoder_stack(d_model, n_head, dropout, n_layers):
    return TransformerDecoder([ModRZTXDecoderLayer(d_model=d_model, dim_feedforward=d_model * 4, nhead=n_head,
                                                   dropout=dropout, activation="relu") for _ in range(n_layers)])
