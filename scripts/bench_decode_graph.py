"""What would a replayed hipGraph of the greedy decode step buy?  The step (tal_greedy_step_fwd, ~35 dependent launches) is
captured once per session at a fixed prefix length and replayed: one host call per step instead of 35.  Timed against the
ordinary launches for one session, and for K sessions on K streams (one host thread replays them round-robin / K threads).
python scripts/bench_decode_graph.py [U] [sessions ...]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import ASRModel, synth
from tal_asrd_amd.system import _GreedySession
dev = torch.device("cuda:0")
m = ASRModel("2x", num_speakers=6008, vocab_size=10000, use_speaker_head=True)
sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
own = m.state_dict()
for k, v in sd.items():
    own[k] = torch.from_numpy(v.copy())
m.load_state_dict(own); m.to(dev)
U = int(sys.argv[1]) if len(sys.argv) > 1 else 32
counts = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8]
n = int(os.environ.get("REPS", "300"))
KMAX = max(counts)
streams = [torch.cuda.Stream() for _ in range(KMAX)]
sessions, graphs = [], []
for k in range(KMAX):
    with torch.cuda.stream(streams[k]):
        win = {"encoder_out": torch.randn(1, 357, 512, device=dev), "encoder_padding_mask": torch.zeros(1, 357, dtype=torch.bool, device=dev)}
        gen_dev = torch.randint(3, 10000, (1024,), device=dev)
        s = _GreedySession(m, gen_dev, 512, sync_mode=2)
        s.set_window(win)
        for _ in range(3): s.step(0, U)
        streams[k].synchronize()
        sessions.append((s, win, gen_dev))
torch.cuda.synchronize()
for k in range(KMAX):
    s = sessions[k][0]
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=streams[k], capture_error_mode="thread_local"):
        s.enqueue(0, U)
    graphs.append(gr)
torch.cuda.synchronize()

def run_plain(k, reps):
    s = sessions[k][0]
    with torch.cuda.stream(streams[k]):
        for _ in range(reps):
            s.enqueue(0, U)
            while not s.ready(50): pass
def run_graph(k, reps):
    with torch.cuda.stream(streams[k]):      # (a replay goes to the CURRENT stream)
        for _ in range(reps):
            graphs[k].replay()
            streams[k].synchronize()

for name, fn in (("launches", run_plain), ("graph replay", run_graph)):
    for K in counts:
        fn(0, 5)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        th = [threading.Thread(target=fn, args=(k, n)) for k in range(K)]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("%-12s U=%d, %d sessions (one thread each): %.3f ms per step and session, %.3f ms per step overall" % (name, U, K, dt / n * 1e3, dt / n / K * 1e3), flush=True)
# one host thread, K graphs in flight
for K in counts:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        for k in range(K):
            with torch.cuda.stream(streams[k]): graphs[k].replay()
        for k in range(K): streams[k].synchronize()
    dt = time.perf_counter() - t0
    print("graph replay U=%d, %d sessions from ONE thread: %.3f ms per round, %.3f ms per step overall" % (U, K, dt / n * 1e3, dt / n / K * 1e3), flush=True)
