"""The reference's call sites run UNEDITED: they hand the model half-precision audio (`audio_x.half()`,
tal/asr/system.py:91-92,285; `x_wav.cuda().half()`, tal/baseline/reconcile.py:78).  The front-end takes the fp16
waveform as it is, widens it exactly and computes in fp32 from there.  Expected values: fixtures recorded from the
reference's own modules on the fp16-rounded waveform (tests/golden/make_golden.py, sections `half`, `flow`)."""
import numpy as np
import pytest
import torch

from tests.conftest import golden, has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]

LOGIT_TOL = 1e-3      # BASELINE.json north_star: logits within 1e-3 fp32


def dev():
    return torch.device("cuda:0")


def _load(model, weights):
    own = model.state_dict()
    for k, v in weights.items():
        own[k] = torch.from_numpy(np.array(v, copy=True))
    model.load_state_dict(own)
    return model.to(dev())


@pytest.fixture(scope="module")
def sd_model(sd_weights):
    from tal_asrd_amd import SDModel
    return _load(SDModel(), sd_weights)


@pytest.fixture(scope="module")
def asr_model(asr_weights):
    from tal_asrd_amd import ASRModel
    return _load(ASRModel("2x", num_speakers=6008, vocab_size=10000, use_speaker_head=True), asr_weights)


@pytest.mark.parametrize("B,L", [(1, 480000), (3, 40123), (2, 4800000)])
def test_half_waveform_equals_its_widened_copy_bit_for_bit(sd_model, B, L):
    """fp16 -> fp32 is exact, so LogMelSpec on the half tensor == LogMelSpec on `.half().float()`: every bit, both
    workgroup shapes (short / long inputs), odd lengths (2-byte aligned rows)."""
    from tal_asrd_amd import synth
    a32 = torch.from_numpy(synth.synth_audio_batch(B, L, 31 + B)).to(dev())
    a16 = a32.half()
    with torch.no_grad():
        m16 = sd_model.logmelspec(a16)
        m32 = sd_model.logmelspec(a16.float())
        assert m16.dtype == torch.float32
        assert torch.equal(m16, m32)
        assert not torch.equal(m16, sd_model.logmelspec(a32))       # (the rounding of the samples is visible)


def test_reconcile_get_speaker_ids_sequence_as_written(sd_model):
    """tal/baseline/reconcile.py:76-85, line for line (model.model -> the SDModel; torchaudio.load -> the synthetic clip)."""
    from tal_asrd_amd import synth
    g = golden("sd_30s_half")
    x_wav = torch.from_numpy(synth.synth_audio_batch(1, int(g["audio_len"]), int(g["audio_seed"])))
    model = sd_model
    # --- reconcile.py:78-85
    x_wav = x_wav.cuda().half()
    with torch.no_grad():
        feat = model.encode(x_wav, None)
        feat_mat = np.matrix(model.spk_embed_proj(
            feat['encoder_out']
        ).detach().cpu().numpy())
        pred_ids = torch.argmax(model.decode(feat), dim=-1)
    ids = pred_ids.squeeze().detach().cpu().numpy().tolist()
    # ---
    np.testing.assert_array_equal(np.asarray(ids), g["ids"][0])
    np.testing.assert_allclose(np.asarray(feat_mat), g["feat"][0], atol=LOGIT_TOL, rtol=0)
    logits = model.decode(feat)
    np.testing.assert_allclose(logits[:, g["logit_rows"]].cpu().numpy(), g["logit_sample"], atol=LOGIT_TOL, rtol=0)
    # the fused form takes the half waveform too
    f2, i2 = model.speaker_ids(x_wav)
    np.testing.assert_array_equal(i2.cpu().numpy(), g["ids"])
    # ... and the un-rounded fixture is NOT what comes out (the cast is performed, not ignored)
    assert np.abs(logits[:, g["logit_rows"]].cpu().numpy() - golden("sd_30s")["logit_sample"]).max() > LOGIT_TOL


def test_generate_force_half_default(asr_model):
    """System.generate with its default force_half=True on the fp32 waveform (tal/asr/system.py:91-95) == the reference
    on the fp16-rounded waveform."""
    from tal_asrd_amd import synth
    from tal_asrd_amd.system import System
    g = golden("flow_generate_beam1_half")
    lens = g["audio_lens"].tolist()
    audio = torch.from_numpy(synth.synth_audio_batch(2, max(lens), int(g["audio_seed"]), lens=lens)).to(dev())
    sys_ = System(asr_model, spk_weight=1.0)
    seqs, spks = sys_.generate(audio, torch.zeros(2, 1, dtype=torch.long, device=dev()), torch.tensor(lens),
                               length=int(g["length"]), beam_size=1, terminate_token=1, force_output=True)
    for i in range(2):
        np.testing.assert_array_equal(seqs[i].numpy(), g["seq_%d" % i])
        np.testing.assert_array_equal(spks[i].argmax(-1).numpy(), g["spk_argmax_%d" % i])
        np.testing.assert_allclose(spks[i][:, ::200].numpy(), g["spk_sample_%d" % i], atol=2e-3, rtol=0)


def test_generate_unaligned_casts_the_waveform_itself(asr_model):
    """generate_unaligned is handed the UN-rounded fp32 waveform and casts it (`audio_x = audio_x.half()`,
    tal/asr/system.py:285) -- the trajectory recorded from the reference on the rounded waveform comes out; and
    `model.encode(audio.half(), lens)` as system.py:290 calls it returns fp32 tensors."""
    from tal_asrd_amd import synth
    from tal_asrd_amd.system import System
    g = golden("flow_unaligned")
    L = int(g["audio_len"])
    audio = torch.from_numpy(synth.synth_audio_batch(1, L, int(g["audio_seed"]))).to(dev())
    enc = asr_model.encode(audio.half(), torch.tensor([L]))
    assert enc["encoder_out"].dtype == torch.float32 and enc["speaker_out"].dtype == torch.float32
    gen, align = System(asr_model).generate_unaligned(audio, torch.ones(1, 1, dtype=torch.long, device=dev()),
                                                      torch.tensor([L]), max_iters=int(g["max_iters"]), stall_patience=25)
    np.testing.assert_array_equal(gen.cpu().numpy(), g["generated"])
    np.testing.assert_array_equal(np.array([int(c[0]) for c, _ in align]), g["chunk_start"])
    np.testing.assert_allclose(np.stack([a.numpy()[0] for _, a in align]), g["attn"], atol=1e-4, rtol=0)


def test_other_dtypes_still_raise(sd_model):
    from tal_asrd_amd._native import NativeError
    with pytest.raises(NativeError):
        sd_model.logmelspec(torch.zeros(1, 16000, dtype=torch.float64, device=dev()))
    with pytest.raises(NativeError):
        sd_model.logmelspec(torch.zeros(1, 16000, dtype=torch.bfloat16, device=dev()))
    with pytest.raises(NativeError):                                    # only the waveform may be half
        sd_model.spk_embed_proj(torch.zeros(1, 4, 1440, dtype=torch.float16, device=dev()))
