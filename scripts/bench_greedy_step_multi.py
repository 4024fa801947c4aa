"""One decode step of G sessions through shared launches (tal_greedy_step_multi_fwd) against G steps on their own launches:
GPU + launch time per merged step, and the host-side cost of the per-token control flow (_UnalignedRun.prepare / consume).
python scripts/bench_greedy_step_multi.py [prefix tokens]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import ASRModel, synth, _native as N
from tal_asrd_amd.system import System, _GreedySession

dev = torch.device("cuda:0")
U = int(sys.argv[1]) if len(sys.argv) > 1 else 32
asr = ASRModel("2x", num_speakers=6008, vocab_size=10000, use_speaker_head=True)
sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in asr.state_dict().items()})
own = asr.state_dict()
for k, v in sd.items():
    own[k] = torch.from_numpy(v.copy())
asr.load_state_dict(own)
asr.to(dev)
L = 120 * 16000
enc = asr.encode(torch.from_numpy(synth.synth_audio_batch(1, L, 7)).to(dev).half(), torch.tensor([L]))
lib = N.lib()
rng = np.random.default_rng(3)
sessions = []
for k in range(16):
    toks = torch.from_numpy(rng.integers(3, 10000, size=U + 8).astype(np.int64)).to(dev)
    s = _GreedySession(asr, toks, 512)
    sl = slice(20 * k, 20 * k + 357)
    s.set_window({"encoder_out": enc["encoder_out"][:, sl].contiguous(), "encoder_padding_mask": enc["encoder_padding_mask"][:, sl].contiguous()})
    sessions.append(s)
reps = 300
for s in sessions[:1]:
    for _ in range(20): s.step(0, U)
    t0 = time.perf_counter()
    for _ in range(reps): s.step(0, U)
    print("one session, own launches, prefix %d: %.3f ms per step" % (U, 1e3 * (time.perf_counter() - t0) / reps), flush=True)
GS = [int(a) for a in os.environ["GS"].split(",")] if os.environ.get("GS") else [1, 2, 4, 8, 12, 16]
for G in GS:
    ctxs = (C.POINTER(N.GreedyCtx) * 16)(*[C.pointer(s.ctx) for s in sessions])
    hs = (C.c_int64 * 16)(*([0] * 16)); ng = (C.c_int64 * 16)(*([U] * 16))
    h = N.stream_handle()
    def step():
        N.check(lib.tal_greedy_step_multi_fwd(ctxs, hs, ng, G, h))
        for s in sessions[:G]:
            while not s.ready(50): pass
    for _ in range(20): step()
    t0 = time.perf_counter()
    for _ in range(reps): step()
    dt = (time.perf_counter() - t0) / reps
    # enqueue-only cost of the call (no wait)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(50): N.check(lib.tal_greedy_step_multi_fwd(ctxs, hs, ng, G, h))
    t_call = (time.perf_counter() - t1) / 50
    torch.cuda.synchronize()
    print("%d sessions, shared launches: %.3f ms per merged step = %.3f ms per token; the C call alone (34 launches enqueued) %.3f ms" % (G, 1e3 * dt, 1e3 * dt / G, 1e3 * t_call), flush=True)

if os.environ.get("GS"):
    sys.exit(0)
# two half-groups on two streams, both steps enqueued by ONE host thread before either result is awaited
s2 = torch.cuda.Stream()
sessB = []
with torch.cuda.stream(s2):
    for k in range(8):
        toks = torch.from_numpy(rng.integers(3, 10000, size=U + 8).astype(np.int64)).to(dev)
        s = _GreedySession(asr, toks, 512)
        sl = slice(30 * k, 30 * k + 357)
        s.set_window({"encoder_out": enc["encoder_out"][:, sl].contiguous(), "encoder_padding_mask": enc["encoder_padding_mask"][:, sl].contiguous()})
        sessB.append(s)
    hB = N.stream_handle()
torch.cuda.synchronize()
for G in (2, 4, 8):
    cA = (C.POINTER(N.GreedyCtx) * 8)(*[C.pointer(s.ctx) for s in sessions[:8]])
    cB = (C.POINTER(N.GreedyCtx) * 8)(*[C.pointer(s.ctx) for s in sessB])
    hs = (C.c_int64 * 8)(*([0] * 8)); ng = (C.c_int64 * 8)(*([U] * 8))
    hA = sessions[0]._stream
    def step2():
        N.check(lib.tal_greedy_step_multi_fwd(cA, hs, ng, G, hA))
        N.check(lib.tal_greedy_step_multi_fwd(cB, hs, ng, G, hB))
        for s in sessions[:G] + sessB[:G]:
            while not s.ready(50): pass
    for _ in range(20): step2()
    t0 = time.perf_counter()
    for _ in range(reps): step2()
    dt = (time.perf_counter() - t0) / reps
    print("2 x %d sessions on two streams, one host thread: %.3f ms per pair of merged steps = %.3f ms per token" % (G, 1e3 * dt, 1e3 * dt / (2 * G)), flush=True)
