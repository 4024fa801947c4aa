"""One greedy decode step (tal_greedy_step_fwd: the whole step as one C call) at a fixed prefix length, repeated.
python scripts/bench_greedy_step.py [U ...]"""
import os, sys, time
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
win = {"encoder_out": torch.randn(1, 357, 512, device=dev), "encoder_padding_mask": torch.zeros(1, 357, dtype=torch.bool, device=dev)}
gen_dev = torch.randint(3, 10000, (1024,), device=dev)
sess = _GreedySession(m, gen_dev, 512)
sess.set_window(win)
n = int(os.environ.get("REPS", "200"))
for U in [int(a) for a in sys.argv[1:]] or [1, 16, 32, 64, 128]:
    for _ in range(10): sess.step(0, U)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): sess.step(0, U)
    torch.cuda.synchronize(); print("U=%d: %.3f ms/step" % (U, (time.perf_counter() - t0) / n * 1e3), flush=True)
