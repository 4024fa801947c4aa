"""Import the reference's own modules (read-only, from /root/reference) so that
golden vectors can be recorded from them.  BUILD-CONTAINER ONLY: nothing under
tests/ imports this at test time and /root/reference does not exist on the GPU
box.  No reference source is copied; the files are executed where they lie.

Third-party packages the reference imports but this image lacks are replaced
by minimal stand-ins *for import purposes only* (they are never the thing under
test), with one exception that is stated loudly everywhere it matters:
torchaudio.transforms.MelSpectrogram is replaced by the oracle's restatement of
torchaudio 0.4.0 (oracle/tal_oracle.py) -- the log-mel front-end is therefore
"parity unpinned" (SURVEY.md section 8c).
"""
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import tal_oracle as O  # noqa: E402


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _load(name, path, package=False):
    spec = importlib.util.spec_from_file_location(
        name, os.path.join(REF, path),
        submodule_search_locations=[os.path.dirname(os.path.join(REF, path))] if package else None)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


class _Spectrogram(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("window", O.hann_window())


class _MelScale(nn.Module):
    def __init__(self, n_mels, sr):
        super().__init__()
        self.register_buffer("fb", O.mel_filterbank(n_mels=n_mels, sr=sr))


class MelSpectrogramStandIn(nn.Module):
    """Oracle restatement of torchaudio 0.4.0 MelSpectrogram: [B, L] -> [B, n_mels, T]."""

    def __init__(self, sample_rate=16000, n_fft=400, win_length=None, hop_length=None, n_mels=128, **kw):
        super().__init__()
        assert (sample_rate, n_fft, win_length, hop_length) == (16000, 400, 400, 160)
        self.spectrogram = _Spectrogram()
        self.mel_scale = _MelScale(n_mels, sample_rate)

    def forward(self, audio):
        p = O.power_spectrogram(audio.float())                 # [B, 201, T]
        return torch.matmul(p.transpose(1, 2), self.mel_scale.fb).transpose(1, 2)


def _legacy_transformer_decoder_forward(self, tgt, memory, tgt_mask=None, memory_mask=None,
                                        tgt_key_padding_mask=None, memory_key_padding_mask=None, **_):
    """torch 1.4 nn.TransformerDecoder.forward: a plain loop over the layers.
    torch >= 2 passes tgt_is_causal/memory_is_causal, which the reference's
    ModRZTXDecoderLayer.forward (tal/asr/models.py:512) does not accept."""
    out = tgt
    for layer in self.layers:
        out = layer(out, memory, tgt_mask=tgt_mask, memory_mask=memory_mask,
                    tgt_key_padding_mask=tgt_key_padding_mask,
                    memory_key_padding_mask=memory_key_padding_mask)
    if self.norm is not None:
        out = self.norm(out)
    return out


_loaded = {}


def load_reference():
    """Returns a namespace with the reference's models / modules / util / uisrnn / system modules."""
    if _loaded:
        return _loaded["ns"]
    nn.TransformerDecoder.forward = _legacy_transformer_decoder_forward

    # third-party stand-ins (import plumbing only)
    _mod("torchaudio", transforms=_mod("torchaudio.transforms", MelSpectrogram=MelSpectrogramStandIn),
         load=None)
    _mod("rezero")
    _mod("rezero.transformer", RZTXDecoderLayer=type("RZTXDecoderLayer", (nn.Module,), {}))
    _mod("fairseq")
    _mod("fairseq.models")
    _mod("fairseq.models.wav2vec", Wav2VecModel=type("Wav2VecModel", (), {}))
    _mod("wandb")
    pl = _mod("pytorch_lightning", LightningModule=nn.Module,
              data_loader=lambda f: f)
    pl.logging = _mod("pytorch_lightning.logging", WandbLogger=object, rank_zero_only=lambda f: f)

    # the reference package is `tal/` but imports itself as `wildspeech`
    ws = _load("wildspeech", "tal/__init__.py", package=True)
    ws.modules = _load("wildspeech.modules", "tal/modules.py")
    ws.optimizers = _load("wildspeech.optimizers", "tal/optimizers.py")
    ws.schedules = _load("wildspeech.schedules", "tal/schedules.py")
    asr = _mod("wildspeech.asr")
    asr.__path__ = [os.path.join(REF, "tal/asr")]
    names = ["ASRAlignedDataset", "ASRAlignedCollater", "RandomSegmentDataset", "AudioCollator",
             "ASRSegmentDataset"]
    asr.data = _mod("wildspeech.asr.data", DEFAULT_SR=16000, **{n: object for n in names})
    tok = _mod("wildspeech.asr.tokenizers")
    tok.sentencepiece = _mod("wildspeech.asr.tokenizers.sentencepiece", Tokenizer=object)
    asr.logger = _mod("wildspeech.asr.logger", WandbLogger=object)
    asr.models = _load("wildspeech.asr.models", "tal/asr/models.py")
    asr.util = _load("wildspeech.asr.util", "tal/asr/util.py")
    asr.system = _load("wildspeech.asr.system", "tal/asr/system.py")
    # transcribe.py: audio file IO / VAD / duration probes are import plumbing only
    # (make_golden feeds the waveform through stand-ins for torchaudio.info / torchaudio.load)
    asr.speech_detect = _mod("wildspeech.asr.speech_detect", get_speech_frames=None)
    _mod("librosa")
    _mod("librosa.core", get_duration=None)
    _mod("mutagen", File=None)
    asr.transcribe = _load("wildspeech.asr.transcribe", "tal/asr/transcribe.py")

    dia = _mod("wildspeech.diarization")
    dia.__path__ = [os.path.join(REF, "tal/diarization")]
    uis_pkg = _mod("wildspeech.diarization.uisrnn")
    uis_pkg.__path__ = [os.path.join(REF, "tal/diarization/uisrnn")]
    for sub in ("loss_func", "utils"):
        setattr(uis_pkg, sub, _load("wildspeech.diarization.uisrnn." + sub,
                                    "tal/diarization/uisrnn/%s.py" % sub))
    uis_pkg.uisrnn = _load("wildspeech.diarization.uisrnn.uisrnn", "tal/diarization/uisrnn/uisrnn.py")

    # the tokenizer base class (decode_speakers, tal/asr/tokenizers/__init__.py:103-138) under its own name: the
    # package name above stays a stand-in because its sentencepiece sub-module needs a model file that does not ship
    tokenizers = _load("wildspeech_asr_tokenizers_base", "tal/asr/tokenizers/__init__.py")

    ns = types.SimpleNamespace(models=asr.models, modules=ws.modules, util=asr.util,
                               system=asr.system, uisrnn=uis_pkg.uisrnn,
                               transcribe=asr.transcribe, tokenizers=tokenizers)
    _loaded["ns"] = ns
    return ns
