#!/usr/bin/env python3
"""Headline benchmark: audio frames/s, waveform-on-device -> log-mel -> TDS encoder ->
diarization head (128-d features + argmax speaker ids), BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W [--workload clip|segments|decode]

One process per GPU.  Typed as above with N > 1 and no WORLD_SIZE in the environment, the script starts its own N
rank processes (127.0.0.1 rendezvous) before anything touches the GPU and prints rank 0's line; under
`python -m torch.distributed.run ... bench.py --gpus N` (how the driver launches it) it reads RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* from the environment.

Workloads (a step = one pass of the hot path over synthetic input already resident in HBM):
  clip      (default at N = 1)  BASELINE.json configs[2]: ONE 1-hour 16 kHz clip per GPU as a single B = 1 call, how the
            reference's whole-episode diarization path consumes audio (tal/baseline/reconcile.py:76-85).  Weak scaling.
  segments  (default at N > 1)  configs[3]: 64 x 5-minute segments = ONE reference call over a [64, L] batch
            (tal/asr/models.py:52: one log-mel mean couples the batch), sharded over the ranks by
            distributed.shard_indices, every rank's share one batched call, the call's global mean restored with one
            (sum, count) scalar all-reduce, ids + features returned to rank 0 through distributed.gather_segments
            (ONE collective per step: ids bit-cast into a 129th feature column, plan built once outside the timed loop)
            inside the timed region.  Strong scaling (the 64 segments are fixed).  At N > 1 rank 0 first runs the SAME
            64 segments alone (`one_gpu_same_workload`), so the line carries its own 1-GPU reference and
            `speedup_vs_one_gpu` -- the default N = 1 line is the 1-hour clip, a different workload.
  decode    configs[4]: the joint decode of a 1-hour episode end to end -- ASR encode + sliding-window greedy decode
            (System.generate_unaligned) + SDModel pass + word-level WDER-format pooling / voting; one episode per GPU.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     -- dominant kernel: algorithmic work / HIP-event time of its launches inside the timed region vs the
                  peak it is bounded by
  cpu_baseline -- the CPU oracle (a port of the reference's PyTorch-CPU path) timed on this box's host cores (N = 1 only):
                  thread count chosen on a 5-minute sample, then the 1-hour clip itself at that count.
The default N = 1 line also carries `per_rank_share` (what one rank of an 8 / 4 / 2 / 1-GPU run of configs[3] computes per
step: 8 / 16 / 32 / 64 five-minute segments as one batched call) and `decode_episode` (configs[4] on a 5-minute episode with
the CPU port of the same chain beside it).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MATRIX_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz
F16_MATRIX_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: BF16/FP16 MFMA dense (v_mfma_f32_32x32x16_f16)
HBM_PEAK_GBS = 8000.0
POINTWISE_MAC_PER_FRAME = 6_272_000   # the 22 pointwise layers of the TDS blocks: fp16x3 form
METRIC = "audio frames/sec (16 kHz, 10 ms hop) end-to-end log-mel -> TDS encoder -> diarization head"
METRIC_DECODE = ("audio frames/sec (16 kHz, 10 ms hop) of the full joint ASR + diarization decode of an episode: ASR encode + "
                 "sliding-window greedy decode + SD pass + WDER-format pooling (NOT the encode-only headline metric)")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["clip", "segments", "decode"], default=None,
                    help="default: clip at --gpus 1, segments (configs[3]) at --gpus > 1")
    ap.add_argument("--seconds", type=float, default=None, help="clip / segment length (clip, decode: 3600; segments: 300)")
    ap.add_argument("--segments", type=int, default=None, help="clip: clips per GPU per step (1); segments: total segments (64)")
    ap.add_argument("--batched", action="store_true", help="clip workload: process a rank's clips as ONE [n, L] call")
    ap.add_argument("--cpu-seconds", type=float, default=300.0, help="clip length of the CPU-baseline thread-count sweep")
    ap.add_argument("--no-cpu-full-clip", action="store_true", help="CPU baseline: the bounded sample only, not the workload's own clip at the best thread count")
    ap.add_argument("--no-per-rank-share", action="store_true", help="clip workload: skip the 8 / 16 / 32 / 64-segment batched calls (per_rank_share)")
    ap.add_argument("--no-decode-episode", action="store_true", help="clip workload: skip the 5-minute configs[4] episode (decode_episode)")
    ap.add_argument("--decode-episode-seconds", type=float, default=300.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="skip the per-launch HIP-event timing")
    ap.add_argument("--no-exact-pass", action="store_true", help="clip workload: skip the extra pass on the exact fp32 kernels (value_exact_f32)")
    ap.add_argument("--no-clip-latency", action="store_true", help="clip workload: skip the 30 s / 5 min call latencies (clip_latency)")
    ap.add_argument("--no-one-gpu-reference", action="store_true", help="segments workload at N > 1: skip rank 0's solo pass over all segments")
    ap.add_argument("--cpu-threads", type=str, default="8,16,32,64,128", help="thread counts the CPU baseline is swept over")
    ap.add_argument("--passes", type=int, default=3, help="timed passes of --steps steps each; `value` is the MEDIAN pass (BASELINE.md section 3: median of >= 3 runs)")
    ap.add_argument("--no-clock-sampler", action="store_true", help="do not sample the shader clock / socket power during the timed passes")
    a = ap.parse_args()
    if a.workload is None:
        a.workload = "clip" if a.gpus == 1 else "segments"
    if a.seconds is None:
        a.seconds = 300.0 if a.workload == "segments" else 3600.0
    if a.segments is None:
        a.segments = 64 if a.workload == "segments" else 1
    return a


def self_launch(args):
    """`python bench.py --gpus N` typed by hand: start the N rank processes (nothing in THIS process has touched the GPU;
    the library is built once here so that the ranks do not race hipcc)."""
    import __graft_entry__ as g
    g.build()
    # this launcher serves the rendezvous store for the whole run, as torch.distributed.run's agent does (the ranks see
    # TORCHELASTIC_USE_AGENT_STORE and connect as clients): the RCCL probe, the agreement on its outcome and every process
    # group of the run hang off ONE store on ONE reserved port.  (A TCP store is host-side only: nothing here touches the GPU.)
    import datetime
    import torch.distributed as dist
    store = dist.TCPStore("127.0.0.1", 0, args.gpus, is_master=True, timeout=datetime.timedelta(seconds=600), wait_for_workers=False)
    port = store.port
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), TORCHELASTIC_USE_AGENT_STORE="True",
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # a rank that dies (before the rendezvous, say) must not leave the others waiting for the collective timeout
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    codes = [None] * len(procs)
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        if any(c not in (None, 0) for c in codes):
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.kill()
                    codes[i] = p.wait()
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    sys.stdout.write("".join(out0))
    sys.stdout.flush()
    return max(abs(c) for c in codes)


_SAMPLER_CHILD = r"""
import sys, time
idx, period = int(sys.argv[1]), float(sys.argv[2])
try:
    import amdsmi
    amdsmi.amdsmi_init()
    hs = amdsmi.amdsmi_get_processor_handles()
    h = hs[idx if idx < len(hs) else 0]
except BaseException as e:
    print("ERR %s: %s" % (type(e).__name__, str(e)[:120]), flush=True)
    sys.exit(0)
def num(v):
    return float(v) if isinstance(v, (int, float)) and 0 < v < 65535 else None
print("OK", flush=True)
import select
while True:
    if select.select([sys.stdin], [], [], period)[0]:
        break                                   # the parent closed (or wrote to) our stdin: done
    try:
        m = amdsmi.amdsmi_get_gpu_metrics_info(h)
        clks = [c for c in (num(v) for v in (m.get("current_gfxclks") or [])) if c is not None]
        sclk = sum(clks) / len(clks) if clks else (num(m.get("current_gfxclk")) or num(m.get("average_gfxclk_frequency")))
        power = num(m.get("current_socket_power")) or num(m.get("average_socket_power"))
        print("%.6f %s %s" % (time.time(), sclk, power), flush=True)
    except BaseException:
        pass
"""


class ClockSampler:
    """Shader clock and socket power of one GPU while the timed passes run, sampled by a CHILD PROCESS (amdsmi: the driver's
    gpu_metrics table; the child never touches HIP, enqueues nothing on the device and -- unlike a thread of this process, which
    round 6 measured at up to +2 % on a pass -- cannot hold this interpreter's lock while a step's launch is due).  Samples carry
    wall-clock stamps; `window(t0, t1)` averages those inside a pass.  The dense layers of the headline step are power-limited
    (1.4-1.9 GHz of 2.4 sustained), so box-to-box and pass-to-pass differences of `value` show up here."""

    def __init__(self, device_index=0, period_s=0.01):
        self.proc, self.why, self.samples = None, None, None
        try:
            self.proc = subprocess.Popen([sys.executable, "-c", _SAMPLER_CHILD, str(int(device_index)), str(period_s)],
                                         stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
            first = self.proc.stdout.readline().strip()
            if first != "OK":
                self.why = first or "the sampler child exited"
                self.close()
        except BaseException as e:      # noqa: B902 -- measurement garnish: never fatal
            self.why = "%s: %s" % (type(e).__name__, str(e)[:120])
            self.proc = None

    def close(self):
        """Stop the child and read what it sampled."""
        if self.proc is None:
            return
        try:
            out, _ = self.proc.communicate(input="\n", timeout=10)
        except BaseException:      # noqa: B902
            self.proc.kill()
            out = ""
        self.proc = None
        self.samples = []
        for ln in out.splitlines():
            p = ln.split()
            if len(p) == 3:
                self.samples.append((float(p[0]), None if p[1] == "None" else float(p[1]), None if p[2] == "None" else float(p[2])))

    def window(self, t0, t1):
        """-> {"sclk_mhz": mean, "sclk_mhz_min": ..., "power_w": mean, "samples": n} of the samples stamped in [t0, t1] (time.time())."""
        if self.samples is None or self.why:
            return {"sclk_mhz": None, "power_w": None, "samples": 0, "unavailable": self.why}
        sc = [s for t, s, _ in self.samples if t0 <= t <= t1 and s is not None]
        pw = [p for t, _, p in self.samples if t0 <= t <= t1 and p is not None]
        return {"sclk_mhz": sum(sc) / len(sc) if sc else None, "sclk_mhz_min": min(sc) if sc else None,
                "power_w": sum(pw) / len(pw) if pw else None, "samples": len(sc)}


def build_sd_model(dev):
    import torch
    from tal_asrd_amd import SDModel, synth
    model = SDModel()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth.fill_state_dict(shapes)
    own = model.state_dict()
    for k, v in sd.items():
        own[k] = torch.from_numpy(v.copy())
    model.load_state_dict(own)
    return model.to(dev), sd


def build_asr_model(dev):
    import torch
    from tal_asrd_amd import ASRModel, synth
    model = ASRModel("2x", num_speakers=6008, vocab_size=10000, use_speaker_head=True)
    sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    own = model.state_dict()
    for k, v in sd.items():
        own[k] = torch.from_numpy(v.copy())
    model.load_state_dict(own)
    return model.to(dev), sd


def dense_layer_algorithmic_bytes(frames, batch=1):
    """Algorithmic HBM bytes of the dense-layer launches of one SD-path step over `batch` clips of `frames` mel frames,
    summed and per launch: every launch reads its activations (+ the residual for the second pointwise layer of a
    TDSBlock) and its weights once and writes its output once.  (For comparison with roofline.traffic.)"""
    def f(t):
        return (t - 21) // 2 + 1
    t1 = f(frames); t2 = f(t1); t3 = f(t2)
    total = 0.0
    launches = 0
    for t, c, depth in ((t1, 800, 2), (t2, 1120, 3), (t3, 1440, 6)):
        act = 4.0 * batch * t * c
        w = 4.0 * c * c
        total += depth * ((act + w + act) + (act + w + act + act))      # relu layer; residual layer
        launches += 2 * depth
    total += 4.0 * (batch * t3 * 1440 + 128 * 1440 + batch * t3 * 128)   # 1440 -> 128 features
    total += 4.0 * (batch * t3 * 128 + 6008 * 128) + 8.0 * batch * t3 * 38   # 128 -> 6008 logits, arg-max partials only
    launches += 2
    return total, launches


# (common.h -- shared declarations, option list -- changes with every other kernel and is deliberately not part of the hash)
DENSE_KERNEL_SOURCES = ("gemm_w64.hip", "gemm_f32.hip", "gemm_s64.hip", "gemm_common.h", "head.hip")


def dense_kernel_sources_sha256():
    """Hash of the sources the dense-layer kernels are built from: a committed roofline.traffic figure (separate rocprofv3
    --pmc passes) is only quoted while it describes THESE kernels (scripts/pmc_traffic_json.py records the same hash)."""
    import hashlib
    h = hashlib.sha256()
    for name in DENSE_KERNEL_SOURCES:
        with open(os.path.join(ROOT, "tal_asrd_amd", "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()


def load_traffic():
    """-> (bytes per dense-layer launch | None, source note).  The newest profiles/rN_pmc_traffic.json whose recorded source
    hash matches the kernels of this tree; a file measured on other kernels is refused (traffic stays null)."""
    import glob
    import re
    want = dense_kernel_sources_sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")),
                   key=lambda p: -int(re.search(r"r(\d+)_pmc_traffic", p).group(1)))
    stale = []
    for path in files:
        try:
            with open(path) as f:
                d = json.load(f)
        except Exception:           # noqa: BLE001
            continue
        name = os.path.basename(path)
        if d.get("kernel_sources_sha256") == want:
            return d["hbm_bytes_per_launch"], ("profiles/%s (builder-run rocprofv3 --pmc passes of the 1-hour clip workload on these "
                                               "kernels -- source hash matches --, not this run)" % name)
        stale.append(name)
    return None, ("none: %s measured other dense-layer kernels than this tree's (source hash differs); re-run "
                  "scripts/profile_round.sh" % (", ".join(stale) or "no profiles/rN_pmc_traffic.json"))


def decode_episode_gpu(system, sdm, audio, lens, sync):
    """BASELINE.json configs[4] on one episode: ASR encode + sliding-window greedy decode (tal/asr/system.py:254-524,654-707)
    + SDModel pass (tal/baseline/reconcile.py:76-85) + word-level WDER-input pooling / voting
    (tal/utils/aligned_to_wder_format.py:150-214).  -> stats dict (ms per leg)."""
    from tal_asrd_amd.wder_format import unaligned_to_wder
    sync()
    t0 = time.perf_counter()
    utts, gen, _ = system.transcribe_unaligned(audio, lens)
    sync(); t1 = time.perf_counter()
    feat, ids = sdm.speaker_ids(audio.half())           # x_wav.cuda().half(), tal/baseline/reconcile.py:78
    sync(); t2 = time.perf_counter()
    tp = feat.shape[1]
    kept = [u for u in utts if int(u["chunkStart"].max()) <= tp - 357]
    ref = [{"episode": "e", "utterance": "x", "speaker": 0, "role": "host"}]
    out = unaligned_to_wder([(ref, kept)], {"e": feat[0]}, {"e": ids[0]}, {}, system.tokenizer, word_level=True, num_ids=6008)
    sync(); t3 = time.perf_counter()
    return dict(tokens=int(gen.shape[1]) - 1, utterances=len(utts), words=len(out[0][1]), decode_ms=1e3 * (t1 - t0),
                sd_ms=1e3 * (t2 - t1), wder_format_ms=1e3 * (t3 - t2), total_ms=1e3 * (t3 - t0)), gen


def decode_episode_cpu(asr_sd, sd, audio_np, threads):
    """The CPU port of the same chain (oracle.generate_unaligned: ASR encode + the reference's per-token full-prefix decode;
    oracle.sd_path: the SDModel pass) on the same episode, at `threads` torch threads.  The WDER-input pooling (44 ms per
    HOUR on the GPU, numpy-sized work) is not part of the port; the GPU figure it is compared with includes it."""
    import torch
    from oracle import tal_oracle as O
    L = audio_np.shape[1]
    before = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        t0 = time.perf_counter()
        toks, _, _ = O.generate_unaligned(audio_np, [[1]], [L], asr_sd)
        t1 = time.perf_counter()
        O.sd_path(audio_np.astype("float16").astype("float32"), sd)
        t2 = time.perf_counter()
    finally:
        torch.set_num_threads(before)
    return {"frames_per_s": (1 + L // 160) / (t2 - t0), "threads": threads, "kind": "port", "seconds": t2 - t0,
            "decode_s": t1 - t0, "sd_s": t2 - t1, "tokens": int(len(toks)) - 1,
            "what": "oracle.generate_unaligned (ASR encode + one full-prefix decoder pass per generated token, as "
                    "tal/asr/system.py:332-411 does) + oracle.sd_path on the same episode, torch-CPU fp32, one run"}, toks


def cpu_baseline(sd, seconds, thread_counts, full_seconds=None):
    """The oracle (CPU restatement of the reference's PyTorch-CPU path) on a bounded sample, swept over thread counts: an
    oversubscribed pool is slower than a smaller one, so the best count is what `value` reports."""
    import torch
    from oracle import tal_oracle as O
    from tal_asrd_amd import synth
    L = int(seconds * 16000)
    audio = synth.synth_audio_batch(1, L, 1234)
    frames = 1 + L // 160
    ncpu = os.cpu_count() or 1
    counts = sorted({min(int(c), ncpu) for c in thread_counts if int(c) > 0}) or [torch.get_num_threads()]
    default_threads = torch.get_num_threads()
    sweep = {}
    try:
        for c in counts:
            torch.set_num_threads(c)
            O.sd_path(audio, sd)                      # warm-up at this pool size
            times = []
            for _ in range(2 if seconds <= 600 else 1):
                t0 = time.perf_counter()
                O.sd_path(audio, sd)
                times.append(time.perf_counter() - t0)
            sweep[c] = frames / min(times)
    finally:
        torch.set_num_threads(default_threads)
    best = max(sweep, key=sweep.get)
    out = {"value": sweep[best], "unit": "frames/s", "cores": best, "kind": "port", "host_cpus": ncpu,
           "threads_swept": {str(k): v for k, v in sweep.items()},
           "sample": "%.0f s synthetic clip (%d frames), torch-CPU fp32, best of %s threads (1 warm-up + best of %d timed runs "
                     "per thread count)%s" % (seconds, frames, counts, 2 if seconds <= 600 else 1,
                                              "" if seconds >= 3600 else "; a bounded sample, not the 1-hour clip `value` is measured on")}
    if full_seconds is not None and full_seconds > seconds:
        # ... and the workload `value` is measured on, at the best thread count (two runs, ~10 s each on the GPU box's host):
        # the 5-minute sample flatters the CPU by 25-40 % (its activations fit the caches better)
        Lf = int(full_seconds * 16000)
        audio_f = synth.synth_audio_batch(1, Lf, 1234)
        ff = 1 + Lf // 160
        torch.set_num_threads(best)
        try:
            times = []
            for _ in range(2):
                t0 = time.perf_counter()
                O.sd_path(audio_f, sd)
                times.append(time.perf_counter() - t0)
        finally:
            torch.set_num_threads(default_threads)
        out["sample_5min"] = {"value": out["value"], "sample": out["sample"]}
        out["value"] = ff / min(times)
        out["sample"] = ("the %.0f s synthetic clip `value` is measured on (%d frames), torch-CPU fp32, %d threads (the best of %s on "
                         "a %.0f s sample), best of 2 runs (%.1f s, %.1f s)" % (full_seconds, ff, best, counts, seconds, times[0], times[1]))
    return out


class FakeSD:
    """TAL_BENCH_FAKE=1 (CPU plumbing self-test of the launch / shard / gather / timing code, tests/test_distributed_cpu.py):
    stands in for the GPU work with deterministic tensors of the right shapes.  Never used for a reported number."""

    def speaker_ids_batch(self, idx, frames):
        import torch
        tp = ((((frames - 21) // 2 + 1) - 21) // 2 + 1 - 21) // 2 + 1
        out = {}
        for i in idx:
            g = torch.Generator().manual_seed(100 + i)
            out[i] = (torch.randn(tp, 128, generator=g), torch.randint(0, 6008, (tp,), generator=g, dtype=torch.int32))
        return out


def open_store(rank, world):
    """The run's rendezvous store: served by the launcher (torch.distributed.run's agent / self_launch) when there is one,
    else by rank 0.  Host-side only."""
    import datetime
    import torch.distributed as dist
    served = os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True"
    return dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")), world,
                         is_master=(rank == 0 and not served), timeout=datetime.timedelta(seconds=300), wait_for_workers=False)


def rccl_probe_child():
    """`bench.py --rccl-probe` (started by rccl_probe as a CHILD process): build an RCCL communicator over all ranks' children
    and run one all-reduce.  Exit code 0 = usable.  TAL_BENCH_RCCL_FAIL simulates the failure modes in the CPU tests:
    "1" raise on every rank, "rankN" raise on rank N only, "hang" never return."""
    rank, world, local = (int(os.environ[k]) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"))
    mode = os.environ.get("TAL_BENCH_RCCL_FAIL", "")
    if mode == "hang":
        time.sleep(3600)
    if mode == "1" or mode == "rank%d" % rank:
        print("TAL_BENCH_RCCL_FAIL is set", file=sys.stderr)
        return 1
    if os.environ.get("TAL_BENCH_FAKE"):
        return 0                      # (plumbing tests: no GPU, nothing to probe on the ranks that are not told to fail)
    import datetime
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    # (always a client: the store is served by the launcher or by rank 0's parent process, which is waiting for this child)
    store = dist.PrefixStore("tal_rccl_probe", dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ["MASTER_PORT"]),
                                                             world, is_master=False, timeout=datetime.timedelta(seconds=30)))
    dist.init_process_group("nccl", store=store, rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=30))
    t = torch.ones(1, device=dev)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    if int(t.item()) != world:
        print("RCCL all_reduce returned %r for %d ranks" % (t.item(), world), file=sys.stderr)
        return 1
    dist.destroy_process_group()
    return 0


def rccl_probe(store, rank, world):
    """Is RCCL usable on this node?  Decided BEFORE this rank touches the GPU, in bounded time, and identically on every rank:
    each rank starts a child process that builds a communicator with the other ranks' children and runs one all-reduce; a
    child that has not finished after TAL_BENCH_PROBE_TIMEOUT (45 s) is killed (a communicator that hangs costs seconds, not
    torch's 300 s watchdog per rank); every rank publishes its outcome in the store and reads all of them, so ONE failing or
    hanging rank sends ALL ranks to gloo.  -> None (RCCL is fine) or the reason it is not."""
    limit = float(os.environ.get("TAL_BENCH_PROBE_TIMEOUT", "45"))
    child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rccl-probe"], stdout=subprocess.DEVNULL,
                             stderr=subprocess.PIPE, text=True)
    try:
        _, err = child.communicate(timeout=limit)
        verdict = "ok" if child.returncode == 0 else "rank %d: %s" % (rank, (err.strip().splitlines() or ["exit code %d" % child.returncode])[-1][:200])
    except subprocess.TimeoutExpired:
        child.kill()                 # (exactly the process started above)
        child.communicate()
        verdict = "rank %d: no communicator + all-reduce within %.0f s" % (rank, limit)
    store.set("tal_rccl_probe_result/%d" % rank, verdict)
    verdicts = [store.get("tal_rccl_probe_result/%d" % r).decode() for r in range(world)]     # (blocks until every rank has published)
    bad = [v for v in verdicts if v != "ok"]
    return bad[0] if bad else None


def main():
    if "--rccl-probe" in sys.argv[1:]:
        raise SystemExit(rccl_probe_child())
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))

    import torch
    import __graft_entry__ as g
    g.build()
    from tal_asrd_amd import _native, synth
    from tal_asrd_amd import distributed as D

    fake = bool(os.environ.get("TAL_BENCH_FAKE"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # TAL_BENCH_BACKEND=gloo is a plumbing self-test only (N ranks sharing the visible GPUs, results
    # staged through host memory); real runs use RCCL ("nccl") with one GPU per rank.
    backend = "gloo" if (fake and not os.environ.get("TAL_BENCH_RCCL_FAIL")) else os.environ.get("TAL_BENCH_BACKEND", "nccl")
    dist = None
    backend_note = None
    store = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        store = open_store(rank, world)
        if backend == "nccl":
            # RCCL first -- probed in child processes before this rank has touched the GPU, under a deadline, with the outcome
            # agreed through the store (rccl_probe): if the communicator cannot be built, its first collective fails or
            # either HANGS on ANY rank, every rank continues on gloo with results staged through host memory and the line SAYS
            # SO (`collective_backend`) -- a degraded number beats none on a node nobody could try beforehand.
            why = rccl_probe(store, rank, world)
            if why is not None:
                backend_note = "gloo (RCCL unusable: %s)" % why
                print("[bench] rank %d: %s" % (rank, backend_note), file=sys.stderr, flush=True)
                backend = "gloo"
    if not fake and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if fake:
        dev = torch.device("cpu")
    else:
        # (fewer visible GPUs than ranks happens in the plumbing tests only: the ranks then share them)
        dev = torch.device("cuda", local_rank % torch.cuda.device_count() if torch.cuda.device_count() < world else local_rank)
        torch.cuda.set_device(dev)
    if world > 1:
        import datetime
        # (every group of the run is namespaced in the one store: no second port, nothing lingers from the probe)
        if backend == "nccl":
            dist.init_process_group("nccl", store=dist.PrefixStore("tal_rccl", store), rank=rank, world_size=world, device_id=dev,
                                    timeout=datetime.timedelta(seconds=180))
        else:
            dist.init_process_group(backend, store=dist.PrefixStore("tal_" + backend, store), rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=600))

    def sync():
        if not fake:
            torch.cuda.synchronize()

    L = int(args.seconds * 16000)
    frames = 1 + L // 160
    lib = None if fake else _native.lib()
    prof = not args.no_prof and not fake
    extra = {}
    exact_mode = bool(lib is not None and _native.get_option("tds_exact_f32"))      # TAL_OPTIONS=tds_exact_f32 runs
    one_gpu_ref = {}

    # ------------------------------------------------------------------ workload set-up
    if args.workload == "decode":
        from tal_asrd_amd.system import System
        from tal_asrd_amd.tokenizer import SynthTokenizer
        asr, asr_sd = build_asr_model(dev)
        sdm, sd = build_sd_model(dev)
        if dist is not None:
            D.broadcast_module(asr)
            D.broadcast_module(sdm)
        system = System(asr, tokenizer=SynthTokenizer(10000))
        audio = torch.from_numpy(synth.synth_audio_batch(1, L, 2469 + rank)).to(dev)     # (generate_unaligned casts it to half, system.py:285)
        lens = torch.tensor([L])
        stats = {}

        def step():
            st, _ = decode_episode_gpu(system, sdm, audio, lens, sync)
            stats.update(st)
        units_per_step = world * frames
        scaling = "weak"
        workload = ("1 x %.0f s 16 kHz episode per GPU: ASRModel encode + sliding-window greedy decode "
                    "(System.generate_unaligned, tal/asr/system.py:254-524) + SDModel pass + word-level WDER-format pooling "
                    "and voting (BASELINE.json configs[4])" % args.seconds)
    elif args.workload == "segments":
        n_seg = args.segments
        lengths = [L] * n_seg
        mine = D.shard_indices(n_seg, rank, world, weights=lengths)
        tp = ((((frames - 21) // 2 + 1) - 21) // 2 + 1 - 21) // 2 + 1          # encoder frames of one segment
        if fake:
            model, sd = FakeSD(), None
        else:
            model, sd = build_sd_model(dev)
            if dist is not None:
                D.broadcast_module(model)     # weights come from rank 0 over RCCL / xGMI (start-up only, one flat buffer)
            from tal_asrd_amd import ops

        def load_batch(idx):
            return torch.cat([torch.from_numpy(synth.synth_audio_batch(1, L, 1234 + i)) for i in idx]).to(dev) if idx else None

        def run_share(idx, batch, reduce_mean):
            """This rank's share `idx` of ONE reference call over the [n_seg, L] batch -> {item: [tp, 129]} (128 features +
            the arg-max id bit-cast into the last column)."""
            if fake:
                res = model.speaker_ids_batch(idx, frames)
                return {i: D.pack_feat_ids(res[i][0], res[i][1]) for i in idx}
            # log-mel without the mean, the call's global (sum, count) over all ranks, subtract, encoder + head
            if batch is not None:
                mel, _, st = ops.logmel(model.logmelspec.plan(), batch, eps=model.logmelspec.eps, subtract_mean=False,
                                        return_stats=True)
            else:
                st = torch.zeros(2, dtype=torch.float64, device=dev)
            mean = D.allreduce_logmel_stats(st) if reduce_mean else (st[0] / st[1]).to(torch.float32).reshape(1)
            out = {}
            if batch is not None:
                # (the call's mean rides in the first resize conv's bias, the encoder output goes to the head in the split form:
                #  SDModel.speaker_ids' own sequence, with the mean of the WHOLE call instead of this rank's share)
                feat, ids = model.speaker_ids_from_logmel(mel, mean)
                for k, i in enumerate(idx):
                    out[i] = D.pack_feat_ids(feat[k], ids[k])
            return out

        batch = None if fake else load_batch(mine)
        # the gather plan (who holds which segment, how many rows) is static: built once, outside the timed loop
        gdev = dev if (fake or backend == "nccl") else torch.device("cpu")
        plan = D.SegmentGather({i: tp for i in mine}, n_seg, trailing=(129,) if mine else None,
                               dtype=torch.float32 if mine else None, device=gdev, dst=0) if dist is not None else None
        gathered = {}

        def step():
            local = run_share(mine, batch, dist is not None)
            if plan is None:
                packed = [local[i] for i in range(n_seg)]
            else:
                if gdev != dev:
                    local = {i: v.to(gdev) for i, v in local.items()}
                packed = plan.gather(local)
            gathered["packed"] = packed
        units_per_step = n_seg * frames
        scaling = "strong"
        workload = ("%d x %.0f s 16 kHz segments = one reference call over a [%d, L] batch, sharded over %d GPU(s) by "
                    "distributed.shard_indices, each rank's share one batched call, one scalar (sum, count) all-reduce for "
                    "the call's log-mel mean, ids + features gathered to rank 0 in one collective per step (BASELINE.json "
                    "configs[3]; SDModel path: log-mel -> TDS 80-800-1120-1440 -> 128-d feat + argmax over 6008 speakers)"
                    % (n_seg, args.seconds, n_seg, world))
    else:
        model, sd = build_sd_model(dev)
        if dist is not None:
            D.broadcast_module(model)
        clips = [torch.from_numpy(synth.synth_audio_batch(1, L, 1234 + rank * args.segments + i)).to(dev)
                 for i in range(args.segments)]
        if args.batched:
            clips = [torch.cat(clips, dim=0)]
        pending = []
        gather_feat = gather_ids = None
        if dist is not None and rank == 0:
            with torch.no_grad():
                f0, _ = model.speaker_ids(clips[0])
            n_rows = f0.shape[-2] * args.segments
            gdev = dev if backend == "nccl" else torch.device("cpu")
            gather_feat = [torch.empty(n_rows, f0.shape[-1], device=gdev) for _ in range(world)]
            gather_ids = [torch.empty(n_rows, dtype=torch.int32, device=gdev) for _ in range(world)]

        def step():
            outs = [model.speaker_ids(clip) for clip in clips]
            if dist is not None:
                feat = torch.cat([o[0].reshape(-1, o[0].shape[-1]) for o in outs])
                ids = torch.cat([o[1].reshape(-1) for o in outs])
                if backend != "nccl":
                    feat, ids = feat.cpu(), ids.cpu()
                # Result gather to rank 0 is asynchronous (RCCL's own stream): it overlaps the next step's
                # compute instead of serialising 23 MB x (N-1) of xGMI traffic behind every step; all
                # pending gathers are waited for inside the timed region.
                pending.append((dist.gather(feat, gather_feat if rank == 0 else None, dst=0, async_op=True),
                                dist.gather(ids, gather_ids if rank == 0 else None, dst=0, async_op=True), feat, ids))
        units_per_step = world * args.segments * frames
        scaling = "weak"
        workload = ((("%d x %.0f s 16 kHz clips per GPU as one batched call " if args.batched else
                      "%d x %.0f s 16 kHz clip per GPU, each a whole-episode B=1 call ") % (args.segments, args.seconds)) +
                    "(BASELINE.json configs[2]; SDModel path of tal/baseline/reconcile.py:76-85: "
                    "log-mel -> TDS 80-800-1120-1440 -> 128-d feat + argmax over 6008 speakers)")

    def drain():
        if args.workload == "clip":
            while pending:
                w1, w2, _, _ = pending.pop(0)
                w1.wait()
                w2.wait()

    # PCIe-inclusive figure (reported beside `value`, never as `value`): one pinned-host -> device copy
    h2d_ms = None
    if rank == 0 and args.workload == "clip":
        host = clips[0].cpu().pin_memory()
        sync()
        t_h = time.perf_counter()
        _tmp = host.to(dev, non_blocking=True)
        sync()
        h2d_ms = 1e3 * (time.perf_counter() - t_h)
        del _tmp, host

    # The same path fed from HOST memory as a stream of clips (the reference's loop over episodes, reconcile.py:96-102): the
    # upload of clip i + 1 runs on a copy stream under the compute of clip i (SDModel.speaker_ids_stream)
    streamed = None
    if rank == 0 and args.workload == "clip" and not fake and world == 1:
        host = clips[0].cpu().pin_memory()
        n_stream = 6
        with torch.no_grad():
            for _ in model.speaker_ids_stream([host] * 2):
                pass
            sync()
            t_s = time.perf_counter()
            first_done = None
            for _ in model.speaker_ids_stream([host] * n_stream):
                if first_done is None:
                    first_done = time.perf_counter() - t_s
            sync()
            dt_s = time.perf_counter() - t_s
        streamed = {"clips": n_stream, "value": n_stream * clips[0].shape[0] * frames / dt_s, "unit": "frames/s",
                    "ms_per_clip": 1e3 * dt_s / n_stream, "ms_per_clip_after_the_first": 1e3 * (dt_s - first_done) / (n_stream - 1),
                    "ms_until_first_result": 1e3 * first_done,
                    "what": "%d clips from pinned host memory (PCIe-inclusive): the first upload is exposed, every later one runs on a copy "
                            "stream under the compute of the clip before it (SDModel.speaker_ids_stream)" % n_stream}
        del host

    # The other two clip lengths BASELINE.json names (configs[0]: 30 s, configs[1]: 5 min), as call latencies beside `value`:
    # they take other kernels than the 1-hour clip (64 x 80 / 128 x 96 dense tiles, 64-step conv tiles, K-sliced head layer)
    clip_latency = None
    if rank == 0 and args.workload == "clip" and not fake and world == 1 and not args.no_clip_latency:
        clip_latency = {}
        with torch.no_grad():
            for sec in (30, 300):
                a = torch.from_numpy(synth.synth_audio_batch(1, sec * 16000, 1234)).to(dev)
                for _ in range(5):
                    model.speaker_ids(a)
                sync()
                t_c = time.perf_counter()
                for _ in range(30):
                    model.speaker_ids(a)
                sync()
                dt_c = (time.perf_counter() - t_c) / 30
                clip_latency["%d s" % sec] = {"ms_per_call": 1e3 * dt_c, "frames_per_s": (1 + sec * 100) / dt_c}
                del a
        clip_latency["what"] = "SDModel.speaker_ids on ONE resident clip of that length, 30 calls back to back (each ends with the " \
                               "range-guard read-back, i.e. a device synchronisation)"

    # What ONE rank of an 8 / 4 / 2 / 1-GPU run of configs[3] (64 x 5-minute segments, tal/asr/transcribe.py:124-162 batches)
    # computes per step: its 8 / 16 / 32 / 64 segments as one batched call.  Measurable on one GPU, and what the 8-GPU target
    # hangs on: expected speed-up at N ranks ~ t(64) / (t(64 / N) + gather).
    per_rank_share = None
    if rank == 0 and args.workload == "clip" and not fake and world == 1 and not args.no_per_rank_share:
        per_rank_share = {}
        with torch.no_grad():
            seg = torch.cat([torch.from_numpy(synth.synth_audio_batch(1, 4800000, 1234 + i)) for i in range(64)]).to(dev)
            for nb in (64, 32, 16, 8):
                a = seg[:nb]
                for _ in range(2):
                    model.speaker_ids(a)
                sync()
                t_c = time.perf_counter()
                for _ in range(4):
                    model.speaker_ids(a)
                sync()
                dt_c = (time.perf_counter() - t_c) / 4
                per_rank_share[str(nb)] = {"ms": 1e3 * dt_c, "frames_per_s": nb * 30001 / dt_c}
            del seg, a
        f64 = per_rank_share["64"]["frames_per_s"]
        for nb in (32, 16, 8):
            per_rank_share[str(nb)]["efficiency_vs_64"] = per_rank_share[str(nb)]["frames_per_s"] / f64
        per_rank_share["expected_speedup"] = {str(64 // nb): per_rank_share["64"]["ms"] / (per_rank_share[str(nb)]["ms"] + 0.25)
                                              for nb in (32, 16, 8)}
        per_rank_share["what"] = ("B x 300 s segments as ONE batched SDModel.speaker_ids call on one GPU (B = 64 / N: the share of a rank "
                                  "of an N-GPU run of configs[3]), 4 calls back to back; expected_speedup[N] = ms(64) / (ms(64 / N) + "
                                  "0.25 ms for the result gather and the scalar all-reduce): an estimate from one-GPU measurements, "
                                  "NOT a multi-GPU measurement")

    # BASELINE.json configs[4] beside the headline: a 5-minute episode through the whole joint chain, and the CPU port of
    # the same chain on the same episode (bounded: ~10 s of CPU time).  --workload decode is the 1-hour form.
    decode_episode = None
    if rank == 0 and args.workload == "clip" and not fake and world == 1 and not args.no_decode_episode:
        from tal_asrd_amd.system import System
        from tal_asrd_amd.tokenizer import SynthTokenizer
        asr, asr_sd = build_asr_model(dev)
        system = System(asr, tokenizer=SynthTokenizer(10000))
        Le = int(args.decode_episode_seconds * 16000)
        ep_np = synth.synth_audio_batch(1, Le, 2469)
        ep = torch.from_numpy(ep_np).to(dev)
        with torch.no_grad():
            decode_episode_gpu(system, model, ep, torch.tensor([Le]), sync)         # warm-up
            st, gen = decode_episode_gpu(system, model, ep, torch.tensor([Le]), sync)
        decode_episode = {"seconds": args.decode_episode_seconds, "tokens": st["tokens"], "ms": st["total_ms"],
                          "ms_per_token": st["decode_ms"] / max(st["tokens"], 1), "frames_per_s": (1 + Le // 160) / (1e-3 * st["total_ms"]),
                          "legs_ms": {k: st[k] for k in ("decode_ms", "sd_ms", "wder_format_ms")},
                          "what": "one %.0f s episode: ASRModel encode + System.generate_unaligned (tal/asr/system.py:254-524) + SDModel "
                                  "pass + word-level WDER-input pooling / voting, second of two runs" % args.decode_episode_seconds}
        if not args.no_cpu_baseline:
            extra["_decode_episode_cpu_args"] = (asr_sd, ep_np, gen[0].cpu().numpy())
        del asr, system, ep

    # ------------------------------------------------------------------ timed region
    def timed_pass(with_prof):
        sync()
        if dist is not None:
            dist.barrier()
        sync()
        if with_prof:
            lib.tal_prof_reset()
            lib.tal_prof_enable(1)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        drain()
        sync()
        if dist is not None:
            dist.barrier()
        sync()
        dt = time.perf_counter() - t0
        if with_prof:
            lib.tal_prof_enable(0)
        return dt

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        drain()
        # `value` comes from passes WITHOUT the per-launch event recording: --passes (>= 3) passes of exactly K steps each, every
        # one bracketed by barrier + synchronize and max-reduced over the ranks; `value` is the MEDIAN pass (one pass is a draw: the
        # step is power-limited and round 5's single pass landed 8 % off the same process's next one).  The roofline numbers come from
        # a further pass of the same K steps with the events (they cost a little host and queue time inside the region they measure).
        sampler = None if (args.no_clock_sampler or fake or rank != 0) else ClockSampler(dev.index or 0)
        pass_s, pass_win = [], []
        for _ in range(max(1, args.passes)):
            w0 = time.time()
            dt_p = timed_pass(False)
            pass_win.append((w0, time.time()))
            if dist is not None:
                t = torch.tensor([dt_p], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt_p = float(t.item())
            pass_s.append(dt_p)
        pass_clk = []
        if sampler is not None:
            sampler.close()
            pass_clk = [sampler.window(a, b) for a, b in pass_win]
        order = sorted(range(len(pass_s)), key=lambda i: pass_s[i])
        med = order[(len(order) - 1) // 2]          # (the lower median for an even count: never an average of two passes)
        elapsed = pass_s[med]
        extra["ms_per_step_passes"] = [1e3 * t / args.steps for t in pass_s]
        extra["passes"] = {"n": len(pass_s), "value_is": "median pass (index %d)" % med,
                           "spread": (max(pass_s) - min(pass_s)) / elapsed}
        if pass_clk:
            extra["gpu_sclk_mhz"] = pass_clk[med]["sclk_mhz"]
            extra["gpu_power_w"] = pass_clk[med]["power_w"]
            extra["passes"]["clocks"] = pass_clk
        if prof and args.workload != "decode":
            elapsed_prof = timed_pass(True)
            extra["ms_per_step_with_launch_events"] = 1e3 * elapsed_prof / args.steps
        if args.workload == "clip" and not fake and not exact_mode and not args.no_exact_pass:
            # the same K steps on the exact fp32-input kernels (v_mfma_f32_32x32x2_f32 everywhere, fp32 activations):
            # the driver-observed figure of the mode whose results are bit-for-bit fmaf chains
            _native.set_option("tds_exact_f32", 1)
            try:
                step(); drain()
                elapsed_exact = timed_pass(False)
            finally:
                _native.set_option("tds_exact_f32", 0)
            if dist is not None:
                t = torch.tensor([elapsed_exact], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                elapsed_exact = float(t.item())
            extra["value_exact_f32"] = units_per_step * args.steps / elapsed_exact
            extra["ms_per_step_exact_f32"] = 1e3 * elapsed_exact / args.steps

    with torch.no_grad():
        if args.workload == "segments" and world > 1 and not args.no_one_gpu_reference:
            # the SAME workload on one GPU: rank 0 alone runs all n_seg segments as one batched call (the other ranks wait),
            # so that a scaling sweep has a same-workload reference (the default N = 1 line is the 1-hour clip)
            if rank == 0:
                all_idx = list(range(n_seg))
                full = None if fake else load_batch(all_idx)
                run_share(all_idx, full, False)
                sync()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    run_share(all_idx, full, False)
                sync()
                dt1 = time.perf_counter() - t0
                del full
                one_gpu_ref.update(one_gpu_same_workload={"value": units_per_step * args.steps / dt1, "unit": "frames/s",
                                                          "ms_per_step": 1e3 * dt1 / args.steps, "n_gpus": 1,
                                                          "what": "rank 0 alone, the same %d segments as ONE batched call, no collective" % n_seg},
                                   speedup_vs_one_gpu=dt1 / elapsed)
            dist.barrier()

    if rank == 0:
        total_frames = units_per_step * args.steps
        line = {
            "metric": METRIC_DECODE if args.workload == "decode" else METRIC, "value": total_frames / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None,
            "dtype": "f32" if exact_mode or args.workload == "decode"
                     else "f32 (dense layers: 3 x f16 MFMA hi/lo split, f32 accumulate)",
            "data": "synthetic",
            "config": {"workload": workload, "frames_per_step": units_per_step, "weights": "synthetic deterministic",
                       "audio_resident_in_hbm": True},
        }
        line.update(extra)
        if world > 1:
            line["collective_backend"] = backend_note or ("rccl" if backend == "nccl" else backend)
        if fake:
            line["data"] = "FAKE (TAL_BENCH_FAKE plumbing self-test: no GPU work, not a measurement)"
        if args.workload == "segments":
            got = gathered.get("packed")
            line["gathered_segments"] = len(got) if got is not None else 0
            line.update(one_gpu_ref)
        if args.workload == "decode":
            line["episode"] = stats
            steps_n = max(stats.get("tokens", 1), 1)
            # dominant cost: the decode step, a chain of ~35 dependent launches that stream the decoder weights
            # (67 MB) and the window's K / V^T (5.8 MB) once: latency-bound, priced against HBM for honesty
            wbytes = 4.0 * (4 * (4 * 512 * 512 + 4 * 512 * 512 + 2 * 512 * 2048) + 10000 * 64 + 64 * 512) + 4.0 * 4 * 2 * 357 * 512
            step_ms = stats["decode_ms"] / steps_n
            line["roofline"] = {"bound": "hbm", "kernel": "decode step (tal_greedy_step_fwd: ~35 dependent launches per generated token)",
                                "achieved": wbytes / (step_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": wbytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                "ms_per_decode_step": step_ms, "algorithmic_bytes_per_step": wbytes,
                                "note": "launch-latency-bound, not bandwidth-bound: ~6 us per dependent launch"}
        if prof and args.workload != "decode":
            ms, n, work = C.c_double(), C.c_int64(), C.c_double()
            kern = {}
            for cls, name in ((0, "gemm_nt_f32"), (1, "gconv_res"), (2, "gconv_s2"), (3, "logmel"), (4, "other")):
                _native.check(lib.tal_prof_collect(cls, C.byref(ms), C.byref(n), C.byref(work)))
                kern[name] = {"ms_total": ms.value, "launches": n.value, "work": work.value}
            gm = kern["gemm_nt_f32"]
            achieved = gm["work"] / (gm["ms_total"] * 1e-3) / 1e12 if gm["ms_total"] > 0 else 0.0
            # fabric-side bytes per launch: NOT measured in this run -- the figure of the builder's separate rocprofv3 --pmc
            # FETCH_SIZE / WRITE_SIZE passes (scripts/profile_round.sh), quoted only while it describes this tree's kernels
            traffic, traffic_source = load_traffic()
            f32_only = exact_mode
            peak = FP32_MATRIX_PEAK_TFLOPS if f32_only else F16_MATRIX_PEAK_TFLOPS
            # MFMA flops actually issued: the pointwise layers run as 3 fp16 MFMAs per fp32 product (hi*hi, hi*lo, lo*hi)
            per_rank_frames = (len(mine) if args.workload == "segments" else args.segments) * frames
            pw = 2.0 * POINTWISE_MAC_PER_FRAME * per_rank_frames * args.steps
            issued = gm["work"] + (0.0 if f32_only else 2.0 * pw)
            nb = len(mine) if args.workload == "segments" else (args.segments if args.batched else 1)
            line["roofline"] = {"bound": "mfma",
                                "kernel": "tal::gemm_w64_kernel (256 x 160 tiles; launches below one round of them: gemm_glds_kernel / "
                                          "gemm_s64_kernel) -- the dense layers; TDS pointwise layers in the fp16x3 form: fp32 "
                                          "products as 3 f16 MFMAs, fp32 accumulate" if not f32_only else
                                          "tal::gemm_glds_kernel (fp32 MFMA dense layer, all epilogues)",
                                "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                                "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_source,
                                "achieved_is": "algorithmic fp32 flops (2 M N K per dense layer) / HIP-event time (rank 0)",
                                "issued_mfma_tflops": issued / (gm["ms_total"] * 1e-3) / 1e12 if gm["ms_total"] > 0 else 0.0,
                                "issued_frac": (issued / (gm["ms_total"] * 1e-3) / 1e12 / peak) if gm["ms_total"] > 0 else 0.0,
                                "fp32_matrix_peak": FP32_MATRIX_PEAK_TFLOPS,
                                "avg_launch_ms": gm["ms_total"] / max(gm["launches"], 1),
                                "launches": gm["launches"],
                                "algorithmic_flops_per_launch": gm["work"] / max(gm["launches"], 1),
                                "algorithmic_bytes_per_launch": (lambda tb: tb[0] / tb[1])(dense_layer_algorithmic_bytes(frames, max(nb, 1)))}
            tot = sum(k["ms_total"] for k in kern.values())
            line["kernel_time_share"] = {k: (v["ms_total"] / tot if tot else 0.0) for k, v in kern.items()}
            line["kernel_ms_per_step"] = {k: v["ms_total"] / args.steps for k, v in kern.items()}
        if streamed is not None:
            streamed["fraction_of_resident_value"] = streamed["value"] / line["value"]
            streamed["steady_state_fraction_of_resident_value"] = line["ms_per_step"] / streamed["ms_per_clip_after_the_first"]
            line["stream_of_clips_from_host"] = streamed
        if clip_latency is not None:
            line["clip_latency"] = clip_latency
        if per_rank_share is not None:
            line["per_rank_share"] = per_rank_share
        if h2d_ms is not None:
            line["h2d_ms_per_clip"] = h2d_ms
            line["value_including_h2d"] = total_frames / (elapsed + 1e-3 * h2d_ms * args.segments * args.steps)
        cpu_args = line.pop("_decode_episode_cpu_args", None)
        if not args.no_cpu_baseline and world == 1 and not fake:
            threads = [c for c in args.cpu_threads.split(",") if c.strip()]
            if args.workload == "decode":
                # the CPU port of the same chain on a BOUNDED sample (a 5-minute episode: ~10 s of CPU time; the 1-hour
                # episode would take minutes), at the thread count that is best for the encoder path
                base = cpu_baseline(sd, args.cpu_seconds, threads)
                Ls = int(min(args.seconds, args.decode_episode_seconds) * 16000)
                ep_s = synth.synth_audio_batch(1, Ls, 2469 + rank)
                port = max((decode_episode_cpu(asr_sd, sd, ep_s, c)[0] for c in sorted({base["cores"], min(16, base["host_cpus"])}) for _ in range(2)),
                           key=lambda r: r["frames_per_s"])       # (the token loop's small GEMMs like fewer threads than the encoder)
                line["cpu_baseline"] = {"value": port["frames_per_s"], "unit": "frames/s", "cores": port["threads"], "kind": "port",
                                        "host_cpus": base["host_cpus"], "tokens": port["tokens"], "decode_s": port["decode_s"], "sd_s": port["sd_s"],
                                        "sample": "%.0f s episode (a bounded sample of the same chain, not the %.0f s episode `value` is "
                                                  "measured on): %s" % (Ls / 16000, args.seconds, port["what"])}
            else:
                line["cpu_baseline"] = cpu_baseline(sd, args.cpu_seconds, threads,
                                                    full_seconds=None if (args.no_cpu_full_clip or args.workload != "clip") else args.seconds)
            line["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
            if decode_episode is not None and cpu_args is not None:
                # best of two runs per thread count: the first one also warms the pool and the allocator
                port, toks = max((decode_episode_cpu(cpu_args[0], sd, cpu_args[1], c)
                                  for c in sorted({line["cpu_baseline"]["cores"], min(16, line["cpu_baseline"]["host_cpus"])}) for _ in range(2)),
                                 key=lambda r: r[0]["frames_per_s"])
                port["what"] = port["what"].replace("one run", "best of two runs")
                same = len(toks) == len(cpu_args[2]) and bool((toks == cpu_args[2]).all())
                decode_episode["cpu_port"] = port
                decode_episode["gpu_over_cpu"] = decode_episode["frames_per_s"] / port["frames_per_s"]
                decode_episode["token_stream_identical_to_cpu_port"] = same
        if decode_episode is not None:
            line["decode_episode"] = decode_episode
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
