"""The log-mel front-end in its two forms on one box: the fast transform on the float64 vector ALU (default, csrc/logmel.hip
logmel_fft_kernel) against the float64 matrix-core DFT of rounds 1-4 (`logmel_mfma`); both against a float64 numpy restatement
computed here (rfft of the windowed, reflect-padded frames; NOT the oracle package: a script is not a test).
python scripts/r5_logmel_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g
g.build()
from tal_asrd_amd import SDModel, synth, ops, _native as N
dev = torch.device("cuda:0")
m = SDModel().to(dev)
lm = m.logmelspec
plan = lm.plan()
win = lm.mel_transform.spectrogram.window.detach().cpu().numpy().astype(np.float64)
fb = lm.mel_transform.mel_scale.fb.detach().cpu().numpy().astype(np.float64)


def truth(x, eps):
    x = x.astype(np.float64)
    xp = np.pad(x, (200, 200), mode="reflect")
    T = 1 + len(x) // 160
    idx = np.arange(400)[None, :] + 160 * np.arange(T)[:, None]
    spec = np.fft.rfft(xp[idx] * win[None, :], axis=1)
    return np.log((spec.real ** 2 + spec.imag ** 2) @ fb + eps)


def timed(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for sec, n in ((10, 200), (30, 200), (60, 200), (300, 100), (3600, 20)):
    L = int(sec * 16000)
    x = torch.from_numpy(synth.synth_audio_batch(1, L, 99)).to(dev)
    res = {}
    for form in ("fft", "mfma"):
        N.set_option("logmel_mfma", 1 if form == "mfma" else 0)
        out = ops.logmel(plan, x, eps=lm.eps, subtract_mean=False)
        us = timed(lambda: ops.logmel(plan, x, eps=lm.eps, subtract_mean=False), n)
        res[form] = (out, us)
    N.set_option("logmel_mfma", 0)
    line = "%5d s clip (%6d frames): fast transform %7.1f us, matrix form %7.1f us (log-mel + mean launches, back to back)" % (
        sec, 1 + L // 160, res["fft"][1], res["mfma"][1])
    d = float((res["fft"][0] - res["mfma"][0]).abs().max())
    same = float((res["fft"][0] == res["mfma"][0]).float().mean())
    line += "; forms differ by at most %.2e (%.4f %% of the values bit-identical)" % (d, 100 * same)
    if win is not None and fb is not None and sec <= 300:
        t = truth(x[0].cpu().numpy(), lm.eps)
        line += "; against float64: fast %.2e, matrix %.2e" % (np.abs(res["fft"][0][0].cpu().numpy() - t).max(), np.abs(res["mfma"][0][0].cpu().numpy() - t).max())
    print(line, flush=True)
# half-precision waveform and a ragged batch through the fast form
xb = torch.from_numpy(synth.synth_audio_batch(3, 16000 * 7 + 123, 5)).to(dev)
a = ops.logmel(plan, xb.half(), eps=lm.eps, subtract_mean=False)
bq = ops.logmel(plan, xb.half().float(), eps=lm.eps, subtract_mean=False)
print("fp16 waveform == the same samples as fp32: %s" % bool(torch.equal(a, bq)))
