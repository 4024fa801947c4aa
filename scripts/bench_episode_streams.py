"""configs[4] as a stream of episodes: System.transcribe_unaligned_many with 1 / 2 / 4 / 8 decode sessions in flight
(one HIP stream, one context and one host thread each, shared weights) against the one-at-a-time loop the reference runs
(tal/asr/system.py:625-742).  Every episode's token stream is compared with its solo run.
python scripts/bench_episode_streams.py [seconds] [episodes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import ASRModel, synth
from tal_asrd_amd.system import System
from tal_asrd_amd.tokenizer import SynthTokenizer

dev = torch.device("cuda:0")
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 3600.0
n_ep = int(sys.argv[2]) if len(sys.argv) > 2 else 8
asr = ASRModel("2x", num_speakers=6008, vocab_size=10000, use_speaker_head=True)
sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in asr.state_dict().items()})
own = asr.state_dict()
for k, v in sd.items():
    own[k] = torch.from_numpy(v.copy())
asr.load_state_dict(own)
asr.to(dev)
system = System(asr, tokenizer=SynthTokenizer(10000))
L = int(seconds * 16000)
frames = 1 + L // 160
eps = []
for k in range(n_ep):
    a = synth.synth_audio_batch(1, L, 2469 + k).astype(np.float16).astype(np.float32)
    eps.append((torch.from_numpy(a).pin_memory(), torch.tensor([L])))
# solo reference: one episode after the other, waveform uploaded first (what a loop over transcribe_unaligned does)
system.transcribe_unaligned(eps[0][0].to(dev), eps[0][1])          # warm-up
torch.cuda.synchronize(); t0 = time.perf_counter()
solo = [system.transcribe_unaligned(a.to(dev), lens) for a, lens in eps]
torch.cuda.synchronize(); t_solo = time.perf_counter() - t0
steps = sum(int(g_.shape[1]) - 1 for _, g_, _ in solo)
print("one at a time: %d episodes x %.0f s, %d decode steps, %.3f s = %.0f frames/s, %.3f ms per step"
      % (n_ep, seconds, steps, t_solo, n_ep * frames / t_solo, 1e3 * t_solo / steps), flush=True)
modes = [(k, 1) for k in (1, 2, 4, 8, 16) if k <= n_ep]
ONLY = os.environ.get("MODES")          # e.g. MODES="4x2,2x16,default": only these (threads x group) modes
# sessions advanced in step through SHARED launches (tal_greedy_step_multi_fwd): (host threads, sessions per group)
modes += [(t, gsz) for t, gsz in ((1, 8), (2, 4), (4, 2), (1, 16), (2, 8), (4, 4), (2, 16)) if gsz <= n_ep]
modes.append((None, None))       # the default split of transcribe_unaligned_many
if ONLY:
    want = set(ONLY.split(","))
    modes = [(k, g_) for k, g_ in modes if ("default" if k is None else "%dx%d" % (k, g_)) in want]
for k, gsz in modes:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    many = system.transcribe_unaligned_many(eps, streams=k, group=gsz)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    # tokens and window starts identical to the solo runs always; attention rows bit-identical with group == 1; sessions that advance in
    # groups decode on the unfolded decoder layer (System.FOLD_GROUP_MAX): rows then equal to rounding (the fold re-associates weights)
    # (the default split from 32 episodes on is two threads x groups of 16)
    gsz_eff = gsz if gsz is not None else (16 if n_ep >= 32 else max(1, min(4, -(-n_ep // 4))))
    unfolded = gsz_eff >= System.FOLD_GROUP_MAX
    same = all(torch.equal(a[1].cpu(), b[1].cpu()) and [int(c[0]) for c, _ in a[2]] == [int(c[0]) for c, _ in b[2]] and
               all((float((x[1] - y[1]).abs().max()) < 1e-5) if unfolded else torch.equal(x[1], y[1]) for x, y in zip(a[2], b[2])) for a, b in zip(solo, many))
    what = "default split (streams=None, group=None)" if k is None else "%2d sessions in flight, own launches" % k if gsz == 1 else "%d thread(s) x groups of %d sessions, shared launches" % (k, gsz)
    print("%s: %.3f s = %.0f frames/s (%.2fx one at a time), %.3f ms per step overall, trajectories identical%s: %s"
          % (what, dt, n_ep * frames / dt, t_solo / dt, 1e3 * dt / steps, " (rows to 1e-5: unfolded layer)" if unfolded else "", same), flush=True)
