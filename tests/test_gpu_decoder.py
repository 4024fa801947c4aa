"""GPU parity of the decoder side (ASRModel.decode / decode_spk, ModRZTXDecoderLayer,
PositionalEncoding, CoreRNN) against golden vectors recorded from the reference and
against the oracle.  pytest -m gpu."""
import numpy as np
import pytest
import torch

from tests.conftest import golden, has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]

LOGIT_TOL = 1e-3
ATTN_TOL = 1e-4   # probabilities; the synthetic decoder has |scores| ~ 1e2-1e3, so fp32 softmax inputs carry ~1e-4 relative noise on any platform (CPU oracle vs CPU reference: 5e-6 abs)


def dev():
    return torch.device("cuda:0")


def _load(model, sd):
    own = model.state_dict()
    for k, v in sd.items():
        assert k in own, k
        own[k] = torch.from_numpy(np.array(v, copy=True))
    model.load_state_dict(own)
    return model.to(dev())


def _fill(module, prefix):
    from tal_asrd_amd import synth
    sd = synth.fill_state_dict({prefix + k: tuple(v.shape) for k, v in module.state_dict().items()})
    return _load(module, {k[len(prefix):]: v for k, v in sd.items()})


@pytest.fixture(scope="module")
def asr_model(asr_weights):
    from tal_asrd_amd import ASRModel
    return _load(ASRModel("2x", num_speakers=6008, vocab_size=10000, use_speaker_head=True), asr_weights)


def test_positional_encoding_golden():
    from tal_asrd_amd import PositionalEncoding
    g = golden("posenc")
    m = PositionalEncoding(64, max_len=32).to(dev())
    # the buffer is evaluated by torch's CPU sin/cos/exp at construction, whose last bit depends on
    # the host's SIMD path (AVX-512 in the build container vs the GPU box's CPU): 1e-6, not bit-exact
    np.testing.assert_allclose(m.pe.cpu().numpy(), g["pe"], atol=2e-6, rtol=0)
    x = torch.from_numpy(g["x"]).to(dev())
    y = m(x)
    np.testing.assert_array_equal(y.cpu().numpy(), (x + m.pe[:5]).cpu().numpy())   # the add itself is exact
    np.testing.assert_allclose(y.cpu().numpy(), g["y"], atol=2e-6, rtol=0)
    with pytest.raises(Exception):
        m(torch.zeros(1, 33, 64, device=dev()))


def test_decoder_layer_small_golden():
    """ModRZTXDecoderLayer(64, 4, 256) with the reference's [U,B,E] contract, all mask combinations."""
    from tal_asrd_amd import ModRZTXDecoderLayer
    from tal_asrd_amd.decoder import causal_mask
    g = golden("declayer_small")
    layer = _fill(ModRZTXDecoderLayer(d_model=64, nhead=4, dim_feedforward=256), "declayer.")
    tgt, mem = torch.from_numpy(g["tgt"]).to(dev()), torch.from_numpy(g["mem"]).to(dev())
    cm = causal_mask(7, dev())
    kpm = torch.from_numpy(g["kpm"]).to(dev())
    for tag, tm, km in (("plain", None, None), ("causal", cm, None), ("kpm", None, kpm), ("causal_kpm", cm, kpm)):
        y = layer(tgt, mem, tgt_mask=tm, memory_key_padding_mask=km)
        np.testing.assert_allclose(y.cpu().numpy(), g["y_" + tag], atol=2e-5, rtol=0, err_msg=tag)
        np.testing.assert_allclose(layer.src_attn_weights.cpu().numpy(), g["w_" + tag], atol=ATTN_TOL, rtol=0)
        assert abs(float(layer.src_attn_weights.sum(-1).mean()) - 1.0) < 1e-5


def _memory(asr_model, S):
    from tal_asrd_amd import synth
    audio = torch.from_numpy(synth.synth_audio_batch(1, 480000, 1234)).to(dev())
    enc = asr_model.encode(audio, torch.tensor([480000]))
    return {"encoder_out": enc["encoder_out"][:, :S].contiguous(),
            "speaker_out": enc["speaker_out"][:, :S].contiguous(),
            "encoder_padding_mask": enc["encoder_padding_mask"][:, :S].contiguous()}


def test_asr_decode_golden(asr_model):
    """Decoder fixtures of SURVEY 8c item 4: U in {1,7,64} x S=357, causal on/off, logits first/last
    row, per-layer src_attn_weights last row, decode_spk last row."""
    g = golden("asr_decode")
    mem = _memory(asr_model, int(g["S"]))
    for U in (1, 7, 64):
        y = torch.from_numpy(g["y_%d" % U]).to(dev())
        for causal in (True, False):
            tag = "U%d_%s" % (U, "causal" if causal else "full")
            logits = asr_model.decode(y, mem, causal_mask=causal)
            assert tuple(logits.shape) == (1, U, 10000)
            np.testing.assert_allclose(logits[:, -1].cpu().numpy(), g["logits_last_" + tag], atol=LOGIT_TOL, rtol=0)
            np.testing.assert_allclose(logits[:, 0].cpu().numpy(), g["logits_first_" + tag], atol=LOGIT_TOL, rtol=0)
            attn = torch.stack([l.src_attn_weights[:, -1] for l in asr_model.decoder.layers], 0)
            np.testing.assert_allclose(attn.cpu().numpy(), g["attn_last_" + tag], atol=ATTN_TOL, rtol=0)
            assert int(logits[0, -1].argmax()) == int(g["logits_last_" + tag][0].argmax())
            spk = asr_model.decode_spk(y, mem, causal_mask=causal)
            np.testing.assert_allclose(spk[:, -1].cpu().numpy(), g["spk_last_" + tag], atol=LOGIT_TOL, rtol=0)


def test_asr_decode_b2_kpm_golden(asr_model):
    from tal_asrd_amd import synth
    g = golden("asr_decode")
    lens = [480000, 400000]
    audio = torch.from_numpy(synth.synth_audio_batch(2, 480000, 1234, lens=lens)).to(dev())
    enc = asr_model.encode(audio, torch.tensor(lens))
    y = torch.from_numpy(g["y_b2"]).to(dev())
    logits = asr_model.decode(y, enc, causal_mask=False)
    np.testing.assert_allclose(logits[:, -1].cpu().numpy(), g["logits_last_b2"], atol=LOGIT_TOL, rtol=0)
    attn = torch.stack([l.src_attn_weights[:, -1] for l in asr_model.decoder.layers], 0)
    np.testing.assert_allclose(attn.cpu().numpy(), g["attn_last_b2"], atol=ATTN_TOL, rtol=0)
    # padded memory positions get exactly zero attention
    mask = enc["encoder_padding_mask"]
    assert float(attn[:, 1][:, mask[1]].abs().max()) == 0.0
    spk = asr_model.decode_spk(y, enc, causal_mask=False)
    np.testing.assert_allclose(spk[:, -1].cpu().numpy(), g["spk_last_b2"], atol=LOGIT_TOL, rtol=0)


def test_last_only_and_kv_cache_equivalence(asr_model):
    """The fast forms used by the decode loops (last-position LM head, cached cross K/V) are
    bit-identical to the full forms."""
    from tal_asrd_amd.decoder import asr_decode, asr_decode_spk
    g = golden("asr_decode")
    mem = _memory(asr_model, 357)
    y = torch.from_numpy(g["y_64"]).to(dev())
    full = asr_model.decode(y, mem, causal_mask=False)
    last = asr_decode(asr_model, y, mem, causal=False, last_only=True)
    np.testing.assert_array_equal(full[:, -1].cpu().numpy(), last.cpu().numpy())
    again = asr_decode(asr_model, y, mem, causal=False, last_only=True)   # second call hits the K/V cache
    np.testing.assert_array_equal(last.cpu().numpy(), again.cpu().numpy())
    s_full = asr_model.decode_spk(y, mem, causal_mask=False)
    s_last = asr_decode_spk(asr_model, y, mem, causal=False, last_only=True)
    np.testing.assert_array_equal(s_full[:, -1].cpu().numpy(), s_last.cpu().numpy())


def test_forward_teacher_forced(asr_model, asr_weights):
    """ASRModel.forward (models.py:291-295) against the oracle end to end on a short ragged batch."""
    from oracle import tal_oracle as O
    from tal_asrd_amd import synth
    lens = [64000, 48000]
    audio = synth.synth_audio_batch(2, 64000, 55, lens=lens)
    y = np.array([[0, 17, 4021, 9999, 1], [0, 5, 6, 7, 8]], dtype=np.int64)
    (lm, spk), enc = asr_model(torch.from_numpy(audio).to(dev()), torch.from_numpy(y).to(dev()), torch.tensor(lens))
    with torch.no_grad():
        oenc = O.asr_encode(audio, asr_weights, lens)
        olm, _ = O.asr_decode(y, oenc, asr_weights, causal_mask=True)
        ospk = O.asr_decode_spk(y, oenc, asr_weights, causal_mask=True)
    np.testing.assert_allclose(enc["encoder_out"].cpu().numpy(), oenc["encoder_out"].numpy(), atol=LOGIT_TOL, rtol=0)
    np.testing.assert_allclose(lm.cpu().numpy(), olm.numpy(), atol=LOGIT_TOL, rtol=0)
    np.testing.assert_allclose(spk.cpu().numpy(), ospk.numpy(), atol=LOGIT_TOL, rtol=0)


def test_log_softmax_and_argmax_rows():
    from tal_asrd_amd import ops
    from tal_asrd_amd.decoder import log_softmax
    g = torch.Generator().manual_seed(3)
    x = torch.randn(5, 16008, generator=g) * 6
    x[2, 100] = x[2, 9000] = x[2].max() + 1.0          # exact tie: first index must win
    y = log_softmax(x.to(dev()))
    np.testing.assert_allclose(y.cpu().numpy(), torch.log_softmax(x, -1).numpy(), atol=2e-5, rtol=0)
    ids = ops.argmax_rows(x.to(dev())).cpu().numpy()
    np.testing.assert_array_equal(ids, x.argmax(-1).numpy())
    assert ids[2] == 100


def test_core_rnn_golden():
    from tal_asrd_amd.uisrnn import CoreRNN
    g = golden("gru")
    rnn = _fill(CoreRNN(256, 512, 1, 256), "corernn.")
    m1, h1 = rnn(torch.from_numpy(g["x1"]).to(dev()), torch.from_numpy(g["h0"]).to(dev()))
    np.testing.assert_allclose(m1.cpu().numpy(), g["m1"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(h1.cpu().numpy(), g["h1"], atol=2e-5, rtol=0)
    m3, h3 = rnn(torch.from_numpy(g["x3"]).to(dev()), None)
    np.testing.assert_allclose(m3.cpu().numpy(), g["m3"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(h3.cpu().numpy(), g["h3"], atol=2e-5, rtol=0)
    g2 = golden("gru_depth2")
    rnn2 = _fill(CoreRNN(256, 512, 2, 256), "corernn2.")
    m, h = rnn2(torch.from_numpy(g2["x3"]).to(dev()), None)
    np.testing.assert_allclose(m.cpu().numpy(), g2["m3"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(h.cpu().numpy(), g2["h3"], atol=2e-5, rtol=0)


@pytest.mark.parametrize("B,In,H", [(1, 256, 512), (5, 256, 512), (16, 256, 512), (17, 256, 512), (64, 256, 512), (100, 256, 512),
                                    (3, 512, 512), (7, 16, 32), (4, 48, 80)])
def test_gru_cell_one_launch_against_float64_and_the_three_launch_form(B, In, H):
    """tal_gru_cell_fwd as ONE launch (gru_step_kernel: both weight matrices streamed once, gate math on the MFMA tile's registers)
    against torch.nn.GRUCell arithmetic in float64 (`tal/diarization/uisrnn/uisrnn.py:27-38`) and against the three-launch form of
    rounds 1-5 (option gru_unfused): same results to fp32 rounding; row counts that are not multiples of the 16-row tile, the
    second GRU layer's square shape, widths that are not multiples of 64."""
    import ctypes as C
    from tal_asrd_amd import _native as N, ops
    g = torch.Generator().manual_seed(B * 1000 + In + H)
    x, h = torch.randn(B, In, generator=g), torch.randn(B, H, generator=g) * 0.7
    w_ih, w_hh = torch.randn(3 * H, In, generator=g) / In ** 0.5, torch.randn(3 * H, H, generator=g) / H ** 0.5
    b_ih, b_hh = torch.randn(3 * H, generator=g) * 0.3, torch.randn(3 * H, generator=g) * 0.3
    gi, gh = x.double() @ w_ih.double().T + b_ih.double(), h.double() @ w_hh.double().T + b_hh.double()
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    want = ((1 - z) * n + z * h.double()).numpy()
    lib = N.lib()
    d = [t.to(dev()).contiguous() for t in (x, h, w_ih, w_hh, b_ih, b_hh)]
    nws = lib.tal_gru_cell_workspace_bytes(B, H)
    ws = ops._ws(nws, dev())
    got = {}
    try:
        for unfused in (0, 1):
            N.set_option("gru_unfused", unfused)
            out = torch.full((B + 1, H), 7.0, device=dev())          # (a guard row behind the last one)
            N.check(lib.tal_gru_cell_fwd(*[N.ptr(t) for t in d[:2]], B, In, H, *[N.ptr(t) for t in d[2:]], N.ptr(out), N.ptr(ws), nws,
                                         N.stream_handle()), "tal_gru_cell_fwd")
            assert bool((out[B] == 7.0).all())
            got[unfused] = out[:B].cpu().numpy()
    finally:
        N.set_option("gru_unfused", 0)
    np.testing.assert_allclose(got[0], want, atol=2e-6, rtol=0)
    np.testing.assert_allclose(got[1], want, atol=2e-6, rtol=0)
    np.testing.assert_allclose(got[0], got[1], atol=2e-6, rtol=0)


VARIANTS = {"1x_spk": dict(model_type="1x", num_speakers=6008, vocab_size=10000, use_speaker_head=True),
            "2x_tok": dict(model_type="2x", num_speakers=6008, vocab_size=10000, use_speaker_head=False),
            # embed_size=0 (tal/asr/models.py:104-117,243-246): no factorised embedding -- the `proj_t == nullptr` arms of the LM head
            # (tal_lm_head_fwd) and of the merged LM-head + pick kernel (lm_pick_body) run against the reference here
            "1x_e0": dict(model_type="1x", num_speakers=6008, vocab_size=10000, use_speaker_head=True, embed_size=0)}


@pytest.mark.parametrize("tag", sorted(VARIANTS))
def test_model_variants_decode_and_greedy_flow(tag):
    """The other model variants of tal/asr/models.py:79-84,103 -- '1x' (d = 256, head dim 64) with the speaker head, '2x'
    with speaker ids as 6008 extra vocabulary tokens, '1x' with embed_size=0 -- against fixtures recorded from the reference
    (tests/golden/make_golden.py, section `variants`): state_dict keys, encoder projection, decode / decode_spk last rows,
    attention rows, and a 150-step System.generate_unaligned trajectory (identical tokens and window starts)."""
    import json
    import os
    from tal_asrd_amd import ASRModel, synth
    from tal_asrd_amd.system import System
    from tests.conftest import GOLDEN
    g = golden("asr_variant_" + tag)
    keys = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))["ASRModel_" + tag]
    m = ASRModel(**VARIANTS[tag])
    own = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert own == {k: tuple(s) for k, s in keys}
    m = _load(m, synth.fill_state_dict(own))
    V = int(g["V"])
    assert m.embedding.weight.shape[0] == V
    audio = torch.from_numpy(synth.synth_audio_batch(1, 480000, 1234)).to(dev())
    enc = m.encode(audio, torch.tensor([480000]))
    np.testing.assert_allclose(enc["encoder_out"][:, torch.from_numpy(g["enc_rows"])].cpu().numpy(), g["encoder_out"],
                               atol=LOGIT_TOL, rtol=0)
    S = int(g["S"])
    mem = {"encoder_out": enc["encoder_out"][:, :S].contiguous(),
           "speaker_out": None if enc["speaker_out"] is None else enc["speaker_out"][:, :S].contiguous(),
           "encoder_padding_mask": enc["encoder_padding_mask"][:, :S].contiguous()}
    for U in (1, 7, 64):
        y = torch.from_numpy(g["y_%d" % U]).to(dev())
        for causal in (True, False):
            t = "U%d_%s" % (U, "causal" if causal else "full")
            logits = m.decode(y, mem, causal_mask=causal)
            assert tuple(logits.shape) == (1, U, V)
            np.testing.assert_allclose(logits[:, -1].cpu().numpy(), g["logits_last_" + t], atol=LOGIT_TOL, rtol=0)
            assert int(logits[0, -1].argmax()) == int(g["logits_last_" + t][0].argmax())
            attn = torch.stack([l.src_attn_weights[:, -1] for l in m.decoder.layers], 0)
            np.testing.assert_allclose(attn.cpu().numpy(), g["attn_last_" + t], atol=ATTN_TOL, rtol=0)
            if VARIANTS[tag]["use_speaker_head"]:
                spk = m.decode_spk(y, mem, causal_mask=causal)
                np.testing.assert_allclose(spk[:, -1].cpu().numpy(), g["spk_last_" + t], atol=LOGIT_TOL, rtol=0)
    L = int(g["flow_len"])
    a = synth.synth_audio_batch(1, L, int(g["flow_seed"])).astype(np.float16).astype(np.float32)
    gen, align = System(m).generate_unaligned(torch.from_numpy(a).to(dev()), torch.ones(1, 1, dtype=torch.long, device=dev()),
                                              torch.tensor([L]), max_iters=int(g["flow_iters"]), stall_patience=25)
    np.testing.assert_array_equal(gen.cpu().numpy(), g["flow_generated"])
    np.testing.assert_array_equal(np.array([int(c[0]) for c, _ in align]), g["flow_chunk_start"])
    np.testing.assert_allclose(np.stack([x.numpy()[0] for _, x in align])[::5], g["flow_attn"], atol=ATTN_TOL, rtol=0)
