"""Does the upload of the next 1-hour waveform (230 MB from pinned host memory, copy stream) hide behind the compute of the
current one (SDModel.speaker_ids, 18 ms)?  compute alone, copy alone, both.  python scripts/bench_h2d_overlap.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import SDModel, synth
dev = torch.device("cuda:0")
m = SDModel()
sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
own = m.state_dict()
for k, v in sd.items():
    own[k] = torch.from_numpy(v.copy())
m.load_state_dict(own)
m.to(dev)
L = 3600 * 16000
host = torch.from_numpy(synth.synth_audio_batch(1, L, 1234)).pin_memory()
x = host.to(dev)
buf = torch.empty_like(x)
cs = torch.cuda.Stream()
n = 6
def sync(): torch.cuda.synchronize()
with torch.no_grad():
    for _ in range(2): m.speaker_ids(x)
    sync(); t0 = time.perf_counter()
    for _ in range(n): m.speaker_ids(x)
    sync(); t_c = (time.perf_counter() - t0) / n
    sync(); t0 = time.perf_counter()
    for _ in range(n):
        with torch.cuda.stream(cs): buf.copy_(host, non_blocking=True)
    sync(); t_h = (time.perf_counter() - t0) / n
    sync(); t0 = time.perf_counter()
    for _ in range(n):
        with torch.cuda.stream(cs): buf.copy_(host, non_blocking=True)
        m.speaker_ids(x)
    sync(); t_b = (time.perf_counter() - t0) / n
    # the same copy cut into 16 pieces
    pieces = 16
    step = L // pieces
    sync(); t0 = time.perf_counter()
    for _ in range(n):
        with torch.cuda.stream(cs):
            for p in range(pieces):
                buf[:, p * step:(p + 1) * step].copy_(host[:, p * step:(p + 1) * step], non_blocking=True)
        m.speaker_ids(x)
    sync(); t_p = (time.perf_counter() - t0) / n
print("HSA_ENABLE_SDMA=%s: compute %.2f ms | copy %.2f ms (%.1f GB/s) | both %.2f ms | both, copy in %d pieces %.2f ms"
      % (os.environ.get("HSA_ENABLE_SDMA", "default"), 1e3 * t_c, 1e3 * t_h, 4 * L / t_h / 1e9, 1e3 * t_b, pieces, 1e3 * t_p))
with torch.no_grad():
    for k in (4, 8, 16):
        for _ in m.speaker_ids_stream([host] * 2):
            pass
        sync(); t0 = time.perf_counter()
        marks = []
        for _ in m.speaker_ids_stream([host] * k):
            marks.append(time.perf_counter() - t0)
        sync(); dt = time.perf_counter() - t0
        print("speaker_ids_stream, %2d clips: %.2f ms per clip overall, %.2f ms per clip after the first (first result after %.2f ms)"
              % (k, 1e3 * dt / k, 1e3 * (dt - marks[0]) / (k - 1), 1e3 * marks[0]))
