"""Call-after-call identity of every kernel class on the hot path, quiet AND with a bandwidth hog on another stream.

Why: for four rounds the arg-max head had a barrier with no wait for the wave's own LDS-DMA loads in front of it; on the 239 k rows of
BASELINE configs[3] a few of the speaker ids of `tal/baseline/reconcile.py:76-85` came out wrong in 14-100 % of calls, and it was
found by ONE failure in ~45 quiet suite runs (profiles/r5_head_lds_dma_race.txt).  A timing race that shows once in 45 quiet runs
shows in seconds when something else is pulling on the memory system: every test here runs its repeat loop a second time while a
host thread keeps device-to-device copies of 256 MB in flight on its own stream (~half of the HBM bandwidth taken, L2 and the
Infinity Cache thrashed, the dispatcher sharing CUs with another queue).  tests/test_isa_sync.py is the static half of this guard.

Results must be BIT-identical from call to call: no kernel on the path uses floating-point atomics; every in-launch merge (split-K
fix-up, key-split attention, LM head + pick, arg-max partials) adds or compares in a fixed order behind tickets.
"""
import contextlib
import ctypes as C
import threading

import numpy as np
import pytest
import torch

from tests.conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def dev():
    return torch.device("cuda:0")


def _load(model, sd):
    own = model.state_dict()
    for k, v in sd.items():
        own[k] = torch.from_numpy(np.array(v, copy=True))
    model.load_state_dict(own)
    return model.to(dev()).eval()


@pytest.fixture(scope="module")
def sd_model(sd_weights):
    from tal_asrd_amd import SDModel
    return _load(SDModel(), sd_weights)


@pytest.fixture(scope="module")
def asr_model(asr_weights):
    from tal_asrd_amd import ASRModel
    return _load(ASRModel("2x", num_speakers=6008, vocab_size=10000, use_speaker_head=True), asr_weights)


@contextlib.contextmanager
def hog(active):
    """While the block runs, a host thread keeps three 256 MB device-to-device copies in flight on a stream of its own."""
    if not active:
        yield {"copies": 0}
        return
    d = dev()
    stream = torch.cuda.Stream(device=d)
    with torch.cuda.stream(stream):
        a = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=d).normal_()
        b = torch.empty_like(a)
    stream.synchronize()
    stop = threading.Event()
    stats = {"copies": 0}

    def run():
        pending = []
        with torch.cuda.device(d), torch.cuda.stream(stream):
            while not stop.is_set():
                b.copy_(a, non_blocking=True)
                a[:1024].add_(1.0)                # (a second, tiny kernel: the hog's queue holds dependent launches too)
                ev = torch.cuda.Event()
                ev.record(stream)
                pending.append(ev)
                stats["copies"] += 1
                if len(pending) >= 3:
                    pending.pop(0).synchronize()
        stream.synchronize()

    t = threading.Thread(target=run, daemon=True)
    t.start()
    try:
        yield stats
    finally:
        stop.set()
        t.join()
        assert stats["copies"] >= 3, "the bandwidth hog never ran beside the test (%d copies)" % stats["copies"]


CONTENDED = [pytest.param(False, id="quiet"), pytest.param(True, id="bandwidth-hog")]


@pytest.mark.parametrize("contended", CONTENDED)
@pytest.mark.parametrize("name,B,sec,n", [("1h", 1, 3600, 30), ("64x5min", 64, 300, 15), ("5min", 1, 300, 100), ("30s", 1, 30, 200)])
def test_sd_path_is_the_same_call_after_call(sd_model, name, B, sec, n, contended):
    """SDModel.speaker_ids (log-mel -> TDS -> head + arg-max; BASELINE configs[2], [3], [1], [0] shapes): features and ids of every
    call equal the first call's bit for bit."""
    from tal_asrd_amd import synth
    x = torch.from_numpy(synth.synth_audio_batch(B, sec * 16000, 4321)).to(dev())
    with torch.no_grad():
        f0, i0 = sd_model.speaker_ids(x)
        f0, i0 = f0.clone(), i0.clone()
        bad_f = bad_i = 0
        with hog(contended):
            for _ in range(n):
                f, i = sd_model.speaker_ids(x)
                bad_f += int(not torch.equal(f, f0))
                bad_i += int(not torch.equal(i, i0))
    assert (bad_f, bad_i) == (0, 0), "%s: of %d calls, features differ from the first call's in %d, ids in %d" % (name, n, bad_f, bad_i)


@pytest.mark.parametrize("contended", CONTENDED)
@pytest.mark.parametrize("name,B,sec,n", [("1h", 1, 3600, 20), ("8x5min", 8, 300, 30), ("30s", 1, 30, 150)])
def test_asr_encode_is_the_same_call_after_call(asr_model, name, B, sec, n, contended):
    """ASRModel.encode on the half-precision waveform the reference's callers hand over: encoder_out and speaker_out bit for bit."""
    from tal_asrd_amd import synth
    x = torch.from_numpy(synth.synth_audio_batch(B, sec * 16000, 99)).to(dev()).half()
    lens = torch.full((B,), sec * 16000, dtype=torch.int64)
    with torch.no_grad():
        e0 = asr_model.encode(x, lens)
        eo, so = e0["encoder_out"].clone(), e0["speaker_out"].clone()
        bad = 0
        with hog(contended):
            for _ in range(n):
                e = asr_model.encode(x, lens)
                bad += int(not (torch.equal(e["encoder_out"], eo) and torch.equal(e["speaker_out"], so)))
    assert bad == 0, "%s: %d of %d calls differ from the first" % (name, bad, n)


@pytest.mark.parametrize("contended", CONTENDED)
def test_decoder_calls_are_the_same_call_after_call(asr_model, contended):
    """ASRModel.decode / decode_spk (tal/asr/models.py:203-289) on prefixes of 1, 7, 64 and 200 tokens, B = 2 with key padding:
    100 calls each, logits bit for bit."""
    from tal_asrd_amd import synth
    x = torch.from_numpy(synth.synth_audio_batch(2, 30 * 16000, 5, lens=[480000, 400000])).to(dev()).half()
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        enc = asr_model.encode(x, torch.tensor([480000, 400000]))
        for U in (1, 7, 64, 200):
            y = torch.randint(3, 10000, (2, U), generator=g).to(dev())
            l0, s0 = asr_model.decode(y, enc).clone(), asr_model.decode_spk(y, enc).clone()
            bad = 0
            with hog(contended):
                for _ in range(100):
                    bad += int(not (torch.equal(asr_model.decode(y, enc), l0) and torch.equal(asr_model.decode_spk(y, enc), s0)))
            assert bad == 0, "prefix %d: %d of 100 calls differ from the first" % (U, bad)


def _sessions(asr_model, prefixes):
    from tal_asrd_amd import synth
    from tal_asrd_amd.system import _GreedySession
    L = 60 * 16000
    audio = torch.from_numpy(synth.synth_audio_batch(1, L, 99)).to(dev())
    enc = asr_model.encode(audio.half(), torch.tensor([L]))
    rng = np.random.default_rng(5)
    out = []
    for k, U in enumerate(prefixes):
        toks = torch.from_numpy(rng.integers(3, 10000, size=U + 8).astype(np.int64)).to(dev())
        s = _GreedySession(asr_model, toks, 512)
        sl = slice(40 * k, 40 * k + 357)
        s.set_window({"encoder_out": enc["encoder_out"][:, sl].contiguous(), "encoder_padding_mask": enc["encoder_padding_mask"][:, sl].contiguous()})
        out.append((s, U))
    return out


@pytest.mark.parametrize("contended", CONTENDED)
def test_decode_steps_are_the_same_step_after_step(asr_model, contended):
    """The greedy decode step (tal/asr/system.py:332-411) in its three launch forms -- the chain of 34 launches, the merged step of
    several sessions, the one-launch form -- repeated on the same states: token, attention row and appended device token of every
    repetition equal the first solo step's.  These are the cross-workgroup hand-offs of the decode side (ticketed merges of the
    key-split attention, the split-K FFN-2, the LM head + pick; the one-launch form's phase barriers)."""
    from tal_asrd_amd import _native as N
    lib = N.lib()
    with torch.no_grad():
        sessions = _sessions(asr_model, (1, 17, 40, 96))
        want = []
        for s, U in sessions:
            tok, row = s.step(0, U)
            want.append((tok, row, int(s.gen_dev[U])))
        ctxs = (C.POINTER(N.GreedyCtx) * 8)(*[C.pointer(s.ctx) for s, _ in sessions])
        hs = (C.c_int64 * 8)(0, 0, 0, 0)
        ng = (C.c_int64 * 8)(*[U for _, U in sessions])
        handle = N.stream_handle()
        bad = {"chain": 0, "merged": 0, "one-launch": 0}
        with hog(contended):
            for _ in range(100):                                   # the launch chain
                for (s, U), (tok, row, app) in zip(sessions, want):
                    s.gen_dev[U] = -1
                    t2, r2 = s.step(0, U)
                    bad["chain"] += int(not (t2 == tok and np.array_equal(r2, row) and int(s.gen_dev[U]) == app))
            for _ in range(100):                                   # four sessions in shared launches
                for s, U in sessions:
                    s.gen_dev[U] = -1
                N.check(lib.tal_greedy_step_multi_fwd(ctxs, hs, ng, len(sessions), handle), "tal_greedy_step_multi_fwd")
                for (s, U), (tok, row, app) in zip(sessions, want):
                    assert s.ready(20000)
                    t2, r2 = s.result()
                    bad["merged"] += int(not (t2 == tok and np.array_equal(r2, row) and int(s.gen_dev[U]) == app))
            try:
                # (the one-launch form is the UNFOLDED layer: its reference is the chain under option decode_no_fold)
                N.set_option("decode_no_fold", 1)
                want_nf = []
                for s, U in sessions:
                    s.gen_dev[U] = -1
                    tok, row = s.step(0, U)
                    want_nf.append((tok, row, int(s.gen_dev[U])))
                N.set_option("decode_persist", 1)
                for _ in range(50):                                # the step as one launch
                    for (s, U), (tok, row, app) in zip(sessions, want_nf):
                        s.gen_dev[U] = -1
                        t2, r2 = s.step(0, U)
                        bad["one-launch"] += int(not (t2 == tok and np.array_equal(r2, row) and int(s.gen_dev[U]) == app))
            finally:
                N.set_option("decode_persist", 0)
                N.set_option("decode_no_fold", 0)
        assert bad == {"chain": 0, "merged": 0, "one-launch": 0}, bad
        for s, _ in sessions:
            assert int(s._tickets.abs().sum()) == 0


@pytest.mark.parametrize("contended", CONTENDED)
def test_an_episode_decodes_to_the_same_tokens_every_time(asr_model, contended):
    """System.generate_unaligned over a 2-minute episode, three times: token stream, window starts and attention rows bit for bit."""
    from tal_asrd_amd import synth
    from tal_asrd_amd.system import System
    L = 120 * 16000
    x = torch.from_numpy(synth.synth_audio_batch(1, L, 2469)).to(dev())
    prime = torch.ones(1, 1, dtype=torch.long, device=dev())
    system = System(asr_model)
    g0, a0 = system.generate_unaligned(x, prime, torch.tensor([L]))
    with hog(contended):
        for _ in range(3):
            g1, a1 = system.generate_unaligned(x, prime, torch.tensor([L]))
            assert torch.equal(g0.cpu(), g1.cpu())
            assert len(a0) == len(a1) and all(int(c0[0]) == int(c1[0]) and torch.equal(r0, r1) for (c0, r0), (c1, r1) in zip(a0, a1))
