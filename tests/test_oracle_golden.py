"""Pin the CPU oracle (oracle/tal_oracle.py) against golden vectors recorded
from the reference's own modules (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import tal_oracle as O
from tal_asrd_amd import synth
from tests.conftest import golden

TOL = 2e-5  # oracle and reference are both torch-CPU fp32; only op grouping differs


def _fill(shapes, prefix):
    sd = synth.fill_state_dict({prefix + k: s for k, s in shapes.items()})
    return {k[len(prefix):]: v for k, v in sd.items()}


def _tds_shapes(sizes, depths, groups, k=21):
    sh = {}
    for i in range(1, len(sizes)):
        p = "blocks.%d." % (i - 1)
        sh[p + "0.weight"] = (sizes[i], sizes[i - 1] // groups, k)
        sh[p + "0.bias"] = (sizes[i],)
        for j in range(depths[i - 1]):
            q = p + "1.%d." % j
            sh[q + "resweight"] = (1,)
            sh[q + "conv.0.weight"] = (sizes[i], sizes[i] // groups, k)
            sh[q + "conv.0.bias"] = (sizes[i],)
            for f in ("fc.0", "fc.3"):
                sh[q + f + ".weight"] = (sizes[i], sizes[i], 1)
                sh[q + f + ".bias"] = (sizes[i],)
    return sh


def test_tds_small():
    g = golden("tds_small")
    # key order must match the reference's state_dict order for name-hash identity only;
    # values depend on names, not order.
    sd = _fill(_tds_shapes([8, 16, 24, 32], [1, 1, 2], 8), "tds_small.")
    y = O.tds_forward(g["x"], sd, prefix="", depths=(1, 1, 2), groups=8)
    assert y.shape == g["y"].shape
    np.testing.assert_allclose(y.numpy(), g["y"], atol=TOL, rtol=0)


def test_tdsblock():
    g = golden("tdsblock")
    sh = {"resweight": (1,), "conv.0.weight": (32, 4, 21), "conv.0.bias": (32,),
          "fc.0.weight": (32, 32, 1), "fc.0.bias": (32,), "fc.3.weight": (32, 32, 1), "fc.3.bias": (32,)}
    sd = _fill(sh, "tdsblock.")
    y = O.tds_block(torch.from_numpy(g["x"]), sd, "", 8)
    np.testing.assert_allclose(y.numpy(), g["y"], atol=TOL, rtol=0)


def test_decoder_layer_small():
    g = golden("declayer_small")
    E, FF = 64, 256
    sh = {"resweight": (1,), "resweight_src": (1,)}
    for a in ("self_attn", "multihead_attn"):
        sh[a + ".in_proj_weight"] = (3 * E, E)
        sh[a + ".in_proj_bias"] = (3 * E,)
        sh[a + ".out_proj.weight"] = (E, E)
        sh[a + ".out_proj.bias"] = (E,)
    sh.update({"linear1.weight": (FF, E), "linear1.bias": (FF,), "linear2.weight": (E, FF), "linear2.bias": (E,)})
    sd = _fill(sh, "declayer.")
    tgt, mem = torch.from_numpy(g["tgt"]), torch.from_numpy(g["mem"])
    causal = O._causal_mask(7)
    kpm = torch.from_numpy(g["kpm"])
    for tag, tm, km in (("plain", None, None), ("causal", causal, None), ("kpm", None, kpm), ("causal_kpm", causal, kpm)):
        y, w = O.decoder_layer(tgt, mem, sd, "", 4, tgt_mask=tm, memory_key_padding_mask=km)
        np.testing.assert_allclose(y.numpy(), g["y_" + tag], atol=TOL, rtol=0)
        np.testing.assert_allclose(w.numpy(), g["w_" + tag], atol=TOL, rtol=0)


def test_positional_encoding():
    g = golden("posenc")
    pe = O.positional_encoding(32, 64)
    np.testing.assert_array_equal(pe.numpy(), g["pe"])
    np.testing.assert_allclose((torch.from_numpy(g["x"]) + pe[:5]).numpy(), g["y"], atol=0, rtol=0)


def _check_sd(name, sd_weights, with_lens, half_audio=False):
    g = golden(name)
    B, L = int(g["batch"]), int(g["audio_len"])
    lens = g["audio_lens"].tolist() if with_lens else None
    audio = synth.synth_audio_batch(B, L, int(g["audio_seed"]), lens=lens)
    if half_audio:          # the reference's call sites cast the waveform to half (reconcile.py:78, system.py:92,285)
        audio = audio.astype(np.float16).astype(np.float32)
    with torch.no_grad():
        mel = O.logmel(audio)
        np.testing.assert_allclose(mel[:, g["mel_rows"]].numpy(), g["mel_sample"], atol=TOL, rtol=0)
        np.testing.assert_allclose(mel.double().abs().sum(dim=(1, 2)).numpy(), g["mel_abs_sum"], rtol=1e-6)
        enc = O.sd_encode_features(mel, sd_weights, lens)
        eo = enc["encoder_out"]
        assert eo.shape[1] == O.tds_total_out_len(O.num_frames(L))
        np.testing.assert_allclose(eo[:, g["enc_rows"]].numpy(), g["enc_sample"], atol=1e-4, rtol=0)
        np.testing.assert_allclose(eo.double().sum(dim=1).numpy(), g["enc_chan_sum"], atol=2e-2, rtol=1e-5)
        logits = O.sd_decode(enc, sd_weights)
        np.testing.assert_allclose(logits[:, g["logit_rows"]].numpy(), g["logit_sample"], atol=1e-3, rtol=0)
        ids = logits.argmax(-1).numpy()
        bad = ids != g["ids"]
        assert not (bad & (g["margin"] > 1e-3)).any()
        assert bad.sum() <= 2
        if with_lens:
            np.testing.assert_array_equal(enc["encoder_padding_mask"].numpy(), g["mask"])
    return g


def test_sd_30s(sd_weights):
    g = _check_sd("sd_30s", sd_weights, False)
    assert g["ids"].shape == (1, 358)


def test_sd_30s_half_audio(sd_weights):
    _check_sd("sd_30s_half", sd_weights, False, half_audio=True)


def test_sd_b2_ragged(sd_weights):
    _check_sd("sd_b2_ragged", sd_weights, True)


def test_sd_b4_one_reference_call(sd_weights):
    g = _check_sd("sd_b4_60s", sd_weights, False)
    assert g["ids"].shape == (4, 733)


def test_sd_5min(sd_weights):
    g = _check_sd("sd_5min", sd_weights, False)
    assert g["ids"].shape == (1, 3733)


def test_asr_encode_b2(asr_weights):
    g = golden("asr_enc_b2")
    lens = g["audio_lens"].tolist()
    audio = synth.synth_audio_batch(2, 480000, 1234, lens=lens)
    with torch.no_grad():
        enc = O.asr_encode(audio, asr_weights, lens)
    r = g["rows"]
    np.testing.assert_allclose(enc["encoder_out"][:, r].numpy(), g["encoder_out"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(enc["speaker_out"][:, r].numpy(), g["speaker_out"], atol=1e-4, rtol=0)
    np.testing.assert_array_equal(enc["encoder_padding_mask"].numpy(), g["mask"])


def test_asr_decode(asr_weights):
    g = golden("asr_decode")
    S = int(g["S"])
    with torch.no_grad():
        audio = synth.synth_audio_batch(1, 480000, 1234)
        enc = O.asr_encode(audio, asr_weights, [480000])
        mem = {"encoder_out": enc["encoder_out"][:, :S], "speaker_out": enc["speaker_out"][:, :S],
               "encoder_padding_mask": enc["encoder_padding_mask"][:, :S]}
        for U in (1, 7, 64):
            for causal in (True, False):
                tag = "U%d_%s" % (U, "causal" if causal else "full")
                logits, attn = O.asr_decode(g["y_%d" % U], mem, asr_weights, causal_mask=causal)
                np.testing.assert_allclose(logits[:, -1].numpy(), g["logits_last_" + tag], atol=1e-3, rtol=0)
                np.testing.assert_allclose(logits[:, 0].numpy(), g["logits_first_" + tag], atol=1e-3, rtol=0)
                a = torch.stack([w[:, -1] for w in attn], 0).numpy()
                np.testing.assert_allclose(a, g["attn_last_" + tag], atol=2e-5, rtol=2e-3)
                spk = O.asr_decode_spk(g["y_%d" % U], mem, asr_weights, causal_mask=causal)
                np.testing.assert_allclose(spk[:, -1].numpy(), g["spk_last_" + tag], atol=1e-3, rtol=0)
        lens = [480000, 400000]
        audio = synth.synth_audio_batch(2, 480000, 1234, lens=lens)
        enc2 = O.asr_encode(audio, asr_weights, lens)
        logits, attn = O.asr_decode(g["y_b2"], enc2, asr_weights, causal_mask=False)
        np.testing.assert_allclose(logits[:, -1].numpy(), g["logits_last_b2"], atol=1e-3, rtol=0)
        a = torch.stack([w[:, -1] for w in attn], 0).numpy()
        np.testing.assert_allclose(a, g["attn_last_b2"], atol=2e-5, rtol=2e-3)
        spk = O.asr_decode_spk(g["y_b2"], enc2, asr_weights, causal_mask=False)
        np.testing.assert_allclose(spk[:, -1].numpy(), g["spk_last_b2"], atol=1e-3, rtol=0)


def test_core_rnn():
    g = golden("gru")
    sh = {"gru.weight_ih_l0": (1536, 256), "gru.weight_hh_l0": (1536, 512), "gru.bias_ih_l0": (1536,),
          "gru.bias_hh_l0": (1536,), "linear_mean1.weight": (512, 512), "linear_mean1.bias": (512,),
          "linear_mean2.weight": (256, 512), "linear_mean2.bias": (256,)}
    sd = _fill(sh, "corernn.")
    m1, h1 = O.core_rnn(g["x1"], g["h0"], sd)
    np.testing.assert_allclose(m1.numpy(), g["m1"], atol=TOL, rtol=0)
    np.testing.assert_allclose(h1.numpy(), g["h1"], atol=TOL, rtol=0)
    m3, h3 = O.core_rnn(g["x3"], None, sd)
    np.testing.assert_allclose(m3.numpy(), g["m3"], atol=TOL, rtol=0)
    np.testing.assert_allclose(h3.numpy(), g["h3"], atol=TOL, rtol=0)
    g2 = golden("gru_depth2")
    sh2 = dict(sh)
    sh2.update({"gru.weight_ih_l1": (1536, 512), "gru.weight_hh_l1": (1536, 512),
                "gru.bias_ih_l1": (1536,), "gru.bias_hh_l1": (1536,)})
    sd2 = _fill(sh2, "corernn2.")
    m, h = O.core_rnn(g2["x3"], None, sd2, depth=2)
    np.testing.assert_allclose(m.numpy(), g2["m3"], atol=TOL, rtol=0)
    np.testing.assert_allclose(h.numpy(), g2["h3"], atol=TOL, rtol=0)


@pytest.mark.parametrize("name", ["flow_unaligned_short", "flow_unaligned"])
def test_generate_unaligned_port(asr_weights, name):
    """The CPU port of System.generate_unaligned (bench.py's configs[4] CPU leg) against the trajectories recorded from the
    reference's own function: token stream, recorded window starts, attention rows."""
    g = golden(name)
    L = int(g["audio_len"])
    audio = synth.synth_audio_batch(1, L, int(g["audio_seed"]))        # un-rounded: the port casts to half as system.py:285 does
    toks, starts, rows = O.generate_unaligned(audio, [[1]], [L], asr_weights, max_iters=int(g["max_iters"]), stall_patience=25)
    np.testing.assert_array_equal(toks, g["generated"][0])
    np.testing.assert_array_equal(starts, g["chunk_start"])
    if "attn" in g.files:
        np.testing.assert_allclose(np.stack(rows), g["attn"], atol=1e-5, rtol=0)
    else:
        assert [r.shape[0] for r in rows] == g["attn_len"].tolist()
        np.testing.assert_allclose(np.concatenate(rows), g["attn_flat"], atol=1e-5, rtol=0)


def test_generate_unaligned_port_with_lm(asr_weights):
    """The port's shallow-fusion branch (tal/asr/system.py:368-384) against the trajectory recorded from the reference's own function
    with the stand-in LM."""
    from tests.golden._lm_standin import StandInLM
    g = golden("flow_unaligned_lm")
    L = int(g["audio_len"])
    audio = synth.synth_audio_batch(1, L, int(g["audio_seed"]))
    with torch.no_grad():
        toks, starts, rows = O.generate_unaligned(audio, [[1]], [L], asr_weights, max_iters=int(g["max_iters"]), stall_patience=25,
                                                  lm=StandInLM().eval(), lm_weight=float(g["lm_weight"]), lm_clamp=9999)
    np.testing.assert_array_equal(toks, g["generated"][0])
    np.testing.assert_array_equal(starts, g["chunk_start"])
    np.testing.assert_allclose(np.stack(rows)[::4], g["attn"], atol=1e-5, rtol=0)
