"""configs[4] timing: whole-episode joint decode (encode + sliding-window greedy decode + SD pass + WDER-format
pooling) on one GPU.  python scripts/bench_episode.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import ASRModel, SDModel, synth
from tal_asrd_amd.system import System
from tal_asrd_amd.tokenizer import SynthTokenizer
from tal_asrd_amd.wder_format import unaligned_to_wder

dev = torch.device("cuda:0")
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 3600.0


def load(m):
    sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
    own = m.state_dict()
    for k, v in sd.items():
        own[k] = torch.from_numpy(v.copy())
    m.load_state_dict(own)
    return m.to(dev)


asr = load(ASRModel("2x", num_speakers=6008, vocab_size=10000, use_speaker_head=True))
sdm = load(SDModel())
L = int(seconds * 16000)
audio = torch.from_numpy(synth.synth_audio_batch(1, L, 2468).astype(np.float16).astype(np.float32)).to(dev)
system = System(asr, tokenizer=SynthTokenizer(10000))
lens = torch.tensor([L])
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    enc = asr.encode(audio, lens)
    torch.cuda.synchronize(); t_enc = time.perf_counter() - t0
    t0 = time.perf_counter()
    utts, gen, align = system.transcribe_unaligned(audio, lens)
    torch.cuda.synchronize(); t_dec = time.perf_counter() - t0
    t0 = time.perf_counter()
    feat, ids = sdm.speaker_ids(audio)
    torch.cuda.synchronize(); t_sd = time.perf_counter() - t0
    t0 = time.perf_counter()
    Tp = feat.shape[1]
    kept = [u for u in utts if int(u["chunkStart"].max()) <= Tp - 357]
    ref = [{"episode": "e", "utterance": "x", "speaker": 0, "role": "host"}]
    out = unaligned_to_wder([(ref, kept)], {"e": feat[0]}, {"e": ids[0]}, {}, system.tokenizer, word_level=True, num_ids=6008)
    torch.cuda.synchronize(); t_pool = time.perf_counter() - t0
    n = gen.shape[1] - 1
    frames = 1 + L // 160
    print("rep %d: %d frames, %d tokens, %d utterances | encode %.1f ms | transcribe_unaligned (encode + %d decode steps) %.1f ms "
          "= %.3f ms/step | SD pass %.1f ms | word-level WDER format %.1f ms | episode frames/s %.0f"
          % (rep, frames, n, len(utts), t_enc * 1e3, n, t_dec * 1e3, (t_dec - t_enc) / max(n, 1) * 1e3, t_sd * 1e3, t_pool * 1e3,
             frames / (t_dec + t_sd + t_pool)), flush=True)
