"""UISRNN.predict (tal/diarization/uisrnn/uisrnn.py:470-583) on the sequences of the `uisrnn_predict` fixture and on a longer one,
with the GRU cell as one launch (default) and as the three launches of rounds 1-5 (option gru_unfused): wall time per observation,
and the cell alone at the beam search's row counts.   python scripts/r6_uisrnn_predict.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import _native as N, ops, synth
from tal_asrd_amd.uisrnn import UISRNN
from tests.test_uisrnn_host import cases, model_args, infer_args, weights, sequence
dev = torch.device("cuda:0")
lib = N.lib()
print("== the cell alone (In 256, H 512), us per call, 300 calls back to back")
for B in (1, 4, 10, 16, 32, 64):
    x, h = torch.randn(B, 256, device=dev), torch.randn(B, 512, device=dev)
    w_ih, w_hh = torch.randn(1536, 256, device=dev) / 16, torch.randn(1536, 512, device=dev) / 22
    b_ih, b_hh = torch.randn(1536, device=dev), torch.randn(1536, device=dev)
    out = torch.empty(B, 512, device=dev)
    nws = lib.tal_gru_cell_workspace_bytes(B, 512)
    ws = ops._ws(nws, dev)
    res = {}
    for unfused in (1, 0):
        N.set_option("gru_unfused", unfused)
        call = lambda: lib.tal_gru_cell_fwd(N.ptr(x), N.ptr(h), B, 256, 512, N.ptr(w_ih), N.ptr(w_hh), N.ptr(b_ih), N.ptr(b_hh), N.ptr(out), N.ptr(ws), nws, N.stream_handle())
        for _ in range(20):
            call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            call()
        torch.cuda.synchronize()
        res[unfused] = 1e6 * (time.perf_counter() - t0) / 300
    print("rows %3d: three launches %6.1f us, one launch %6.1f us" % (B, res[1], res[0]))
N.set_option("gru_unfused", 0)
print("== UISRNN.predict, ms per observation (second of two runs)")
for case in cases():
    seq = sequence(case)
    for reps in (1, 8):
        s = np.tile(seq, (reps, 1))
        res = {}
        for unfused in (1, 0):
            N.set_option("gru_unfused", unfused)
            m = UISRNN(model_args(case), device="cuda:0")
            own = m.rnn_model.state_dict()
            for k, v in weights(case).items():
                own[k] = torch.from_numpy(np.array(v, copy=True))
            m.rnn_model.load_state_dict(own)
            m.rnn_model.to("cuda:0")
            m.predict(s, infer_args(case))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pred = m.predict(s, infer_args(case))
            torch.cuda.synchronize()
            res[unfused] = (1e3 * (time.perf_counter() - t0) / (len(s) * case["test_iteration"]), [int(c) for c in pred])
        assert res[0][1] == res[1][1]
        if reps == 1:
            assert res[0][1] == case["pred"]
        print("%-28s x%d (%4d observations, beam %2d, depth %d): three launches %.3f ms, one launch %.3f ms per observation; labels identical"
              % (case["name"], reps, len(s), case["beam_size"], case["depth"], res[1][0], res[0][0]))
N.set_option("gru_unfused", 0)
