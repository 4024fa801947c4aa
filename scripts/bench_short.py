"""Short-clip latency: SDModel.speaker_ids on one clip of the given length, repeated -- eagerly and as the
captured HIP graph of the same launches.  python scripts/bench_short.py [seconds ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import SDModel, synth, ops
dev = torch.device("cuda:0")
m = SDModel()
sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
own = m.state_dict()
for k, v in sd.items():
    own[k] = torch.from_numpy(v.copy())
m.load_state_dict(own); m.to(dev)
n = int(os.environ.get("REPS", "50"))
for sec in [float(a) for a in sys.argv[1:]] or [30.0, 300.0]:
    L = int(sec * 16000)
    x = torch.from_numpy(synth.synth_audio_batch(1, L, 1234)).to(dev)
    with torch.no_grad():
        for _ in range(5): m.speaker_ids(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): m.speaker_ids(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        static_x = x.clone()
        enc = m.encoder
        heads = (m.spk_embed_proj.weight, m.spk_embed_proj.bias, m.spk_logit_proj.weight, m.spk_logit_proj.bias)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            mel, mean = m.logmelspec.forward_unsubtracted(static_x)
            y, chk = ops.tds_forward(enc._descriptor(0, len(enc.sizes) - 1), mel, enc.sizes[-1], defer=True, x_mean=mean, out_split=True)
            feat, _, ids = ops.sd_head(y, *heads, want_logits=False, want_ids=True, x_split=chk.y_split, w_embed_split=m._embed_split())
        def replay():
            static_x.copy_(x); graph.replay()
            assert not chk.flagged()
            return feat.clone(), ids.clone()
        for _ in range(5): replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): replay()
        torch.cuda.synchronize(); dg = (time.perf_counter() - t0) / n
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): graph.replay()
        torch.cuda.synchronize(); dr = (time.perf_counter() - t0) / n
    print("%.0f s clip: eager %.3f ms per call = %.2f M frames/s; as a captured HIP graph (copy in, replay, status read, copies "
          "out) %.3f ms; replays alone back to back %.3f ms" % (sec, dt * 1e3, (1 + L // 160) / dt / 1e6, dg * 1e3, dr * 1e3), flush=True)
