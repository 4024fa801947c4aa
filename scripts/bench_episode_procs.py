"""Does a decode session per PROCESS scale where a session per thread does not?  N worker processes (own HIP runtime, own
copy of the weights) share the one GPU, each decoding `per` episodes of `seconds` seconds; the parent times the whole batch.
python scripts/bench_episode_procs.py [seconds] [episodes per worker] [worker counts ...]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--worker":
    seconds, per, rank, go = float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    import numpy as np
    import torch
    from tal_asrd_amd import ASRModel, synth
    from tal_asrd_amd.system import System
    from tal_asrd_amd.tokenizer import SynthTokenizer
    dev = torch.device("cuda:0")
    asr = ASRModel("2x", num_speakers=6008, vocab_size=10000, use_speaker_head=True)
    sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in asr.state_dict().items()})
    own = asr.state_dict()
    for k, v in sd.items():
        own[k] = torch.from_numpy(v.copy())
    asr.load_state_dict(own)
    asr.to(dev)
    system = System(asr, tokenizer=SynthTokenizer(10000))
    L = int(seconds * 16000)
    eps = [torch.from_numpy(synth.synth_audio_batch(1, L, 2469 + rank * per + k).astype(np.float16).astype(np.float32)).pin_memory()
           for k in range(per)]
    system.transcribe_unaligned(eps[0][:, :16000 * 60].to(dev), torch.tensor([16000 * 60]))      # warm-up
    torch.cuda.synchronize()
    open(go + ".ready%d" % rank, "w").close()
    while not os.path.exists(go):
        time.sleep(0.001)
    t0 = time.perf_counter()
    steps = 0
    for a in eps:
        _, g, _ = system.transcribe_unaligned(a.to(dev, non_blocking=True), torch.tensor([L]))
        steps += int(g.shape[1]) - 1
    torch.cuda.synchronize()
    print("%d %d %.6f" % (rank, steps, time.perf_counter() - t0), flush=True)
    sys.exit(0)

import __graft_entry__ as g
g.build()
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 900.0
per = int(sys.argv[2]) if len(sys.argv) > 2 else 2
counts = [int(x) for x in sys.argv[3:]] or [1, 2, 4, 8]
frames = 1 + int(seconds * 16000) // 160
for n in counts:
    go = "/tmp/tal_go_%d_%d" % (os.getpid(), n)
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(seconds), str(per), str(r), go],
                              stdout=subprocess.PIPE, text=True) for r in range(n)]
    while not all(os.path.exists(go + ".ready%d" % r) for r in range(n)):
        time.sleep(0.01)
        if any(p.poll() not in (None, 0) for p in procs):
            raise SystemExit("a worker died")
    t0 = time.perf_counter()
    open(go, "w").close()
    outs = [p.communicate()[0] for p in procs]
    dt = time.perf_counter() - t0
    steps = sum(int(o.split()[1]) for o in outs)
    slow = max(float(o.split()[2]) for o in outs)
    print("%2d processes x %d episodes of %.0f s: %.3f s wall (slowest worker %.3f s), %d steps -> %.0f frames/s, %.3f ms per step overall"
          % (n, per, seconds, dt, slow, steps, n * per * frames / slow, 1e3 * slow / steps), flush=True)
    for r in range(n):
        os.remove(go + ".ready%d" % r)
    os.remove(go)
