"""Decode steps under a kernel trace: MODE=solo (one session, prefix 20), merged8 (eight sessions in shared launches, uniform 20-token
prefixes) or mixed8 (prefixes 12 43 20 8 31 25 16 37).  rocprofv3 --kernel-trace -- python3 scripts/r5_merged_step_trace.py;
scripts/rocpd_summary.py on the result gives the per-kernel averages (the warm-up steps are in it too: same kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import ASRModel, synth, _native as N
from tal_asrd_amd.system import _GreedySession

dev = torch.device("cuda:0")
mode = os.environ.get("MODE", "solo")
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
for k in range(8):
    toks = torch.from_numpy(rng.integers(3, 10000, size=200).astype(np.int64)).to(dev)
    s = _GreedySession(asr, toks, 512)
    sl = slice(20 * k, 20 * k + 357)
    s.set_window({"encoder_out": enc["encoder_out"][:, sl].contiguous(), "encoder_padding_mask": enc["encoder_padding_mask"][:, sl].contiguous()})
    sessions.append(s)
torch.cuda.synchronize()
pre = [20] * 8 if mode != "mixed8" else [12, 43, 20, 8, 31, 25, 16, 37]
G = 1 if mode == "solo" else 8
ctxs = (C.POINTER(N.GreedyCtx) * 16)(*[C.pointer(s.ctx) for s in sessions])
hs = (C.c_int64 * 16)(*([0] * 16)); ng = (C.c_int64 * 16)(*(pre + [0] * 8))
h = N.stream_handle()
for _ in range(int(os.environ.get("REPS", "40"))):
    if G == 1:
        sessions[0].step(0, pre[0])
    else:
        N.check(lib.tal_greedy_step_multi_fwd(ctxs, hs, ng, G, h))
        for s in sessions[:G]:
            while not s.ready(50): pass
torch.cuda.synchronize()
print("done", mode)
