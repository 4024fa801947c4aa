"""T host threads x G sessions per merged decode step (tal_greedy_step_multi_fwd), each thread on a stream of its own hardware queue
(tal_asrd_amd.hwqueues): does a chain of merged launches slow down when other chains run beside it, and where -- in the call that
enqueues the 34 launches or in the wait for the result?  Uniform and mixed prefix lengths.
python scripts/r5_threads_step.py"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import ASRModel, synth, hwqueues, _native as N
from tal_asrd_amd.system import _GreedySession

dev = torch.device("cuda:0")
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
    toks = torch.from_numpy(rng.integers(3, 10000, size=200).astype(np.int64)).to(dev)
    s = _GreedySession(asr, toks, 512)
    sl = slice(20 * k, 20 * k + 357)
    s.set_window({"encoder_out": enc["encoder_out"][:, sl].contiguous(), "encoder_padding_mask": enc["encoder_padding_mask"][:, sl].contiguous()})
    sessions.append(s)
torch.cuda.synchronize()
streams = hwqueues.spread(dev, 4)      # (leased for the life of this script)
REPS = int(os.environ.get("REPS", "400"))


def run(T, G, prefixes):
    """-> per-thread (ms per merged step, ms in the launch call, ms waiting)"""
    out = [None] * T
    start = threading.Barrier(T)

    def work(t):
        mine = sessions[t * G:(t + 1) * G]
        ctxs = (C.POINTER(N.GreedyCtx) * 16)(*[C.pointer(s.ctx) for s in mine])
        hs = (C.c_int64 * 16)(*([0] * 16))
        ng = (C.c_int64 * 16)(*[prefixes[(t * G + i) % len(prefixes)] for i in range(G)] + [0] * (16 - G))
        with torch.cuda.stream(streams[t]):
            h = N.stream_handle()
            tl = tp = 0.0
            def step():
                nonlocal tl, tp
                a = time.perf_counter()
                rc = lib.tal_greedy_step_multi_fwd(ctxs, hs, ng, G, h) if G > 1 else lib.tal_greedy_step_fwd(mine[0]._ctx_ref, 0, ng[0], 3, h)
                b = time.perf_counter()
                assert rc == 0
                for s in mine:
                    while not s.ready(50): pass
                c = time.perf_counter()
                tl += b - a; tp += c - b
            for _ in range(30): step()
            start.wait()
            tl = tp = 0.0
            t0 = time.perf_counter()
            for _ in range(REPS): step()
            dt = time.perf_counter() - t0
            out[t] = (1e3 * dt / REPS, 1e3 * tl / REPS, 1e3 * tp / REPS)
    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    for x in th: x.start()
    for x in th: x.join()
    return out


MIX = [12, 43, 20, 8, 31, 25, 16, 37]
for name, pre in (("uniform 32-token prefixes", [32]), ("uniform 20-token prefixes", [20]), ("mixed prefixes %s" % MIX, MIX)):
    print("# %s" % name, flush=True)
    for T, G in ((1, 1), (1, 2), (2, 2), (4, 2), (1, 4), (2, 4), (4, 4), (1, 8), (2, 8)):
        if T * G > 16:
            continue
        r = run(T, G, pre)
        ms = max(x[0] for x in r)
        print("%d thread(s) x %d sessions per merged step: %.3f ms per step (slowest thread; launch call %.3f, wait %.3f) = %.4f ms per token overall" %
              (T, G, ms, np.mean([x[1] for x in r]), np.mean([x[2] for x in r]), ms / (T * G)), flush=True)
