"""GPU parity of the decode control flow (tal_asrd_amd.system.System.generate /
generate_unaligned) against trajectories recorded from the reference's own
tal/asr/system.py functions (tests/golden/make_golden.py, section `flow`):
identical token ids, identical window trajectory, identical speaker-change (EOS) indices."""
import numpy as np
import pytest
import torch

from tests.conftest import golden, has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def asr_model(asr_weights):
    from tal_asrd_amd import ASRModel
    m = ASRModel("2x", num_speakers=6008, vocab_size=10000, use_speaker_head=True)
    own = m.state_dict()
    for k, v in asr_weights.items():
        own[k] = torch.from_numpy(np.array(v, copy=True))
    m.load_state_dict(own)
    return m.to(dev())


def test_generate_beam1_with_speaker_head(asr_model):
    from tal_asrd_amd import synth
    from tal_asrd_amd.system import System
    g = golden("flow_generate_beam1")
    lens = g["audio_lens"].tolist()
    audio = torch.from_numpy(synth.synth_audio_batch(2, max(lens), int(g["audio_seed"]), lens=lens)).to(dev())
    sys_ = System(asr_model, spk_weight=1.0)
    seqs, spks = sys_.generate(audio, torch.zeros(2, 1, dtype=torch.long, device=dev()), torch.tensor(lens),
                               length=int(g["length"]), beam_size=1, terminate_token=1, force_half=False,
                               force_output=True)
    for i in range(2):
        np.testing.assert_array_equal(seqs[i].numpy(), g["seq_%d" % i])
        np.testing.assert_array_equal(spks[i].argmax(-1).numpy(), g["spk_argmax_%d" % i])
        np.testing.assert_allclose(spks[i][:, ::200].numpy(), g["spk_sample_%d" % i], atol=2e-3, rtol=0)


def test_generate_beam3_terminates(asr_model):
    from tal_asrd_amd import synth
    from tal_asrd_amd.system import System
    g = golden("flow_generate_beam3")
    lens = g["audio_lens"].tolist()
    audio = torch.from_numpy(synth.synth_audio_batch(2, max(lens), int(g["audio_seed"]), lens=lens)).to(dev())
    sys_ = System(asr_model, spk_weight=0.0)
    seqs, spks = sys_.generate(audio, torch.zeros(2, 1, dtype=torch.long, device=dev()), torch.tensor(lens),
                               length=int(g["length"]), beam_size=3, terminate_token=int(g["terminate_token"]),
                               force_half=False, force_output=False)
    for i in range(2):
        want = g["seq_%d" % i]
        if want.size == 0:
            assert seqs[i] is None
        else:
            np.testing.assert_array_equal(seqs[i].numpy(), want)
        assert spks[i] is None


@pytest.mark.parametrize("beam", [1, 3])
def test_generate_with_lm_shallow_fusion(asr_model, beam):
    """The LM branch of System.generate (tal/asr/system.py:127-138: last-position LM log-probabilities, speaker tokens clamped away,
    added with lm_weight on the shared vocabulary) against fixtures recorded from the reference's own function with a stand-in LM
    (the reference ships no LM class): beam 1 with the speaker head and force_output, beam 3 with a terminate token."""
    from tal_asrd_amd import synth
    from tal_asrd_amd.system import System
    from tal_asrd_amd.tokenizer import SynthTokenizer
    from tests.golden._lm_standin import StandInLM
    g = golden("flow_generate_lm_beam%d" % beam)
    lens = g["audio_lens"].tolist()
    audio = torch.from_numpy(synth.synth_audio_batch(2, max(lens), int(g["audio_seed"]), lens=lens)).to(dev())
    lm = StandInLM().eval().to(dev())
    sys_ = System(asr_model, spk_weight=1.0 if beam == 1 else 0.0, tokenizer=SynthTokenizer(10000), lm=lm, lm_weight=float(g["lm_weight"]))
    seqs, spks = sys_.generate(audio, torch.zeros(2, 1, dtype=torch.long, device=dev()), torch.tensor(lens), length=int(g["length"]),
                               beam_size=beam, terminate_token=int(g["terminate_token"]), force_half=False, force_output=(beam == 1))
    for i in range(2):
        want = g["seq_%d" % i]
        if want.size == 0:
            assert seqs[i] is None
        else:
            np.testing.assert_array_equal(seqs[i].numpy(), want)
        if beam == 1:
            np.testing.assert_array_equal(spks[i].argmax(-1).numpy(), g["spk_argmax_%d" % i])
    # without the LM the decode is a different one (the fixture pins the fusion, not the plain path)
    if beam == 1:
        plain = golden("flow_generate_beam1")
        assert (g["seq_0"] != plain["seq_0"]).any()


def test_generate_unaligned_trajectory(asr_model):
    from tal_asrd_amd import synth
    from tal_asrd_amd.system import System
    from tal_asrd_amd.util import split_speaker_turns
    g = golden("flow_unaligned")
    L = int(g["audio_len"])
    audio = synth.synth_audio_batch(1, L, int(g["audio_seed"])).astype(np.float16).astype(np.float32)
    sys_ = System(asr_model)
    gen, align = sys_.generate_unaligned(torch.from_numpy(audio).to(dev()),
                                         torch.ones(1, 1, dtype=torch.long, device=dev()), torch.tensor([L]),
                                         max_iters=int(g["max_iters"]), stall_patience=25)
    np.testing.assert_array_equal(gen.cpu().numpy(), g["generated"])
    np.testing.assert_array_equal(np.array([int(c[0]) for c, _ in align]), g["chunk_start"])
    got_attn = np.stack([a.numpy()[0] for _, a in align])
    np.testing.assert_allclose(got_attn, g["attn"], atol=1e-4, rtol=0)
    # speaker-change (EOS) indices are a pure function of the token stream -> identical
    want_turns = split_speaker_turns(g["generated"][0].tolist(), 10000)
    got_turns = split_speaker_turns(gen[0].cpu().tolist(), 10000)
    assert got_turns == want_turns


def test_generate_unaligned_with_lm_shallow_fusion(asr_model):
    """The LM branch of System.generate_unaligned (tal/asr/system.py:368-384) against a trajectory recorded from the reference's own
    function with the stand-in LM: per step the LM sees the live prefix (speaker tokens clamped), lm_weight x its last-position
    log-probabilities reach the fused LM-head + pick kernel as an additive row (tal_greedy_ctx.pick_bias).  Tokens and window
    starts identical (closest recorded decision 2.7e-3), attention rows 1e-4; 163 of the 261 tokens differ from the run without
    the LM, so the fixture pins the fusion."""
    from tal_asrd_amd import synth
    from tal_asrd_amd.system import System
    from tal_asrd_amd.tokenizer import SynthTokenizer
    from tests.golden._lm_standin import StandInLM
    g = golden("flow_unaligned_lm")
    L = int(g["audio_len"])
    audio = synth.synth_audio_batch(1, L, int(g["audio_seed"])).astype(np.float16).astype(np.float32)
    lm = StandInLM().eval().to(dev())
    sys_ = System(asr_model, tokenizer=SynthTokenizer(10000), lm=lm, lm_weight=float(g["lm_weight"]))
    x = torch.from_numpy(audio).to(dev())
    prime = torch.ones(1, 1, dtype=torch.long, device=dev())
    gen, align = sys_.generate_unaligned(x, prime, torch.tensor([L]), max_iters=int(g["max_iters"]), stall_patience=25)
    np.testing.assert_array_equal(gen.cpu().numpy(), g["generated"])
    np.testing.assert_array_equal(np.array([int(c[0]) for c, _ in align]), g["chunk_start"])
    np.testing.assert_allclose(np.stack([a.numpy()[0] for _, a in align])[::4], g["attn"], atol=1e-4, rtol=0)
    plain = golden("flow_unaligned")["generated"]
    n = min(plain.shape[1], g["generated"].shape[1])
    assert (g["generated"][0, :n] != plain[0, :n]).sum() > 50
    # a weight of 0 is the plain decode (the branch is skipped as in the reference: `self.args.lm_weight > 0`)
    gen0, _ = System(asr_model, tokenizer=SynthTokenizer(10000), lm=lm, lm_weight=0.0).generate_unaligned(
        x, prime, torch.tensor([L]), max_iters=40, stall_patience=25)
    genp, _ = System(asr_model).generate_unaligned(x, prime, torch.tensor([L]), max_iters=40, stall_patience=25)
    np.testing.assert_array_equal(gen0.cpu().numpy(), genp.cpu().numpy())
    # several episodes in flight with an LM: one launch chain per session (no merged steps), same trajectory as the solo run
    solo, _ = sys_.generate_unaligned(x, prime, torch.tensor([L]), max_iters=60, stall_patience=25)
    outs = sys_.transcribe_unaligned_many([(x, torch.tensor([L]))] * 2, streams=2, group=4, max_iters=60, stall_patience=25)
    for _, gen2, _ in outs:
        np.testing.assert_array_equal(gen2.cpu().numpy(), solo.cpu().numpy())
    assert (solo.cpu().numpy()[0, :41] != genp.cpu().numpy()[0]).any()

def test_generate_unaligned_episode_shorter_than_the_window(asr_model):
    """20 s episode = 233 encoder frames < the 357-frame window: the window start is clamped to a NEGATIVE value and the
    reference's python slices wrap (tal/asr/system.py:347-348,480); windows of 233 and then 124 frames, recorded values
    0 and -124.  Tokens, recorded window starts and the (ragged) attention rows match the reference's run."""
    from tal_asrd_amd import synth
    from tal_asrd_amd.system import System
    g = golden("flow_unaligned_short")
    L = int(g["audio_len"])
    audio = synth.synth_audio_batch(1, L, int(g["audio_seed"])).astype(np.float16).astype(np.float32)
    gen, align = System(asr_model).generate_unaligned(torch.from_numpy(audio).to(dev()),
                                                      torch.ones(1, 1, dtype=torch.long, device=dev()), torch.tensor([L]),
                                                      max_iters=int(g["max_iters"]), stall_patience=25)
    np.testing.assert_array_equal(gen.cpu().numpy(), g["generated"])
    np.testing.assert_array_equal(np.array([int(c[0]) for c, _ in align]), g["chunk_start"])
    assert [a.shape[1] for _, a in align] == g["attn_len"].tolist()
    np.testing.assert_allclose(np.concatenate([a.numpy()[0] for _, a in align]), g["attn_flat"], atol=1e-4, rtol=0)


def test_beam_topk_kernel_matches_torch():
    from tal_asrd_amd.system import _beam_topk
    g = torch.Generator().manual_seed(11)
    B, beam, V = 3, 4, 10000
    lp = torch.log_softmax(torch.randn(B * beam, V, generator=g) * 3, -1)
    sc = torch.randn(B * beam, generator=g)
    done = torch.zeros(B * beam, dtype=torch.uint8)
    done[5] = 1
    total = lp + sc.view(-1, 1)
    total[5] = float("-inf")
    want_v, want_i = torch.topk(total.view(B, beam * V), k=beam)
    v, i = _beam_topk(lp.to(dev()), sc.to(dev()), done.to(dev()), B, beam, beam)
    np.testing.assert_array_equal(i.cpu().numpy(), want_i.numpy())
    np.testing.assert_array_equal(v.cpu().numpy(), want_v.numpy())


def test_windowed_transcription_matches_reference_transcribe_file(asr_model):
    """tal/asr/transcribe.py:79-169 recorded from the reference's own transcribe_file (windows of 10 s,
    7.5 s apart, batches of 2, a window whose beam never finishes is dropped)."""
    import json
    import os
    from tal_asrd_amd import synth
    from tal_asrd_amd.system import System
    from tal_asrd_amd.transcribe import transcribe_file
    with open(os.path.join(os.path.dirname(__file__), "golden", "flow_transcribe.json")) as f:
        g = json.load(f)
    audio = torch.from_numpy(synth.synth_audio_batch(1, g["audio_len"], g["audio_seed"])[0]).to(dev())
    sys_ = System(asr_model, spk_weight=0.0)

    def text(seq):
        return " ".join(str(int(t)) for t in seq)
    for beam in (1, 2):
        got = transcribe_file(audio, sys_, g["window"], g["stride"], batch_size=g["batch_size"], beam_width=beam,
                              length=g["length"], use_eot=True, eot_token_id=g["eot"], decode=text, force_half=False)
        assert got == g["beam%d_texts" % beam]
        spliced = transcribe_file(audio, sys_, g["window"], g["stride"], batch_size=g["batch_size"],
                                  beam_width=beam, length=g["length"], use_eot=True, eot_token_id=g["eot"],
                                  decode=text, splice=True, force_half=False)
        assert spliced == g["beam%d_spliced" % beam]
    raw = transcribe_file(audio, sys_, g["window"], g["stride"], batch_size=3, beam_width=1, length=g["length"],
                          use_eot=True, eot_token_id=g["eot"], force_half=False)
    assert [text(s) for s in raw] == g["beam1_texts"]       # batch composition does not change results
