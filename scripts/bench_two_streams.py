"""Two independent SD calls side by side on two HIP streams against the same calls one after the other: does the chip do
better when one call's grouped convs (issue-bound, far below the power limit) run beside the other's dense layers
(power-limited)?  python scripts/bench_two_streams.py [seconds per clip] [calls per stream]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import SDModel, synth
dev = torch.device("cuda:0")
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 1800.0
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 10
m = SDModel()
sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
own = m.state_dict()
for k, v in sd.items():
    own[k] = torch.from_numpy(v.copy())
m.load_state_dict(own); m.to(dev)
L = int(seconds * 16000)
clips = [torch.from_numpy(synth.synth_audio_batch(1, L, 40 + k)).to(dev) for k in range(2)]
frames = 1 + L // 160
with torch.no_grad():
    ref = [m.speaker_ids(c) for c in clips]
    for _ in range(2):
        for c in clips: m.speaker_ids(c)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(calls):
        for c in clips: m.speaker_ids(c)
    torch.cuda.synchronize(); t_seq = time.perf_counter() - t0
print("one after the other: %d x 2 clips of %.0f s: %.2f ms per pair = %.2f M frames/s" % (calls, seconds, 1e3 * t_seq / calls, 2 * calls * frames / t_seq / 1e6), flush=True)

def run(k, stream, out, delay):
    with torch.no_grad(), torch.cuda.stream(stream):
        if delay: time.sleep(delay)
        for _ in range(calls):
            out[k] = m.speaker_ids(clips[k])
        stream.synchronize()

for delay in (0.0, 0.004, 0.008):
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    out = [None, None]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(k, streams[k], out, delay * k)) for k in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); t_par = time.perf_counter() - t0
    same = all(torch.equal(out[k][1], ref[k][1]) and torch.equal(out[k][0], ref[k][0]) for k in range(2))
    print("side by side on two streams (second one %.0f ms late): %.2f ms per pair = %.2f M frames/s (%.3fx), results identical: %s"
          % (1e3 * delay, 1e3 * t_par / calls, 2 * calls * frames / t_par / 1e6, t_seq / t_par, same), flush=True)
