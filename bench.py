#!/usr/bin/env python3
"""Headline benchmark: audio frames/s, waveform-on-device -> log-mel -> TDS encoder ->
diarization head (128-d features + argmax speaker ids), BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (launched by torch.distributed.run for N > 1, rank/addr from the
env).  A step = one pass of the hot path over this rank's clip(s) already resident in
HBM.  Default workload (config[2] of BASELINE.json): ONE 1-hour 16 kHz synthetic clip per
GPU processed as a single B=1 call, exactly how the reference's whole-episode
diarization path consumes audio (tal/baseline/reconcile.py:76-85).  Weak scaling: every
rank processes its own clip (independent episodes; no data-path collective except the
result gather of ids + features to rank 0, which is inside the timed region).

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     -- dominant kernel (the dense layers): algorithmic fp32 flops / HIP-event time of its launches
                  inside the timed region vs the matrix peak of the MFMA it issues (fp16: 2500 TFLOP/s; the TDS
                  pointwise layers run as 3 f16 MFMAs per fp32 product; TAL_TDS_F32=1: pure fp32, 157.3)
  cpu_baseline -- the CPU oracle (a port of the reference's PyTorch-CPU path) timed on
                  this box's host cores on a bounded 5-minute sample.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MATRIX_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz
F16_MATRIX_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: BF16/FP16 MFMA dense (v_mfma_f32_32x32x16_f16)
POINTWISE_MAC_PER_FRAME = 6_272_000   # the 22 pointwise layers of the TDS blocks: fp16x3 form
MAC_PER_FRAME_GEMM = 6_272_000 + 119_168 / 1.0  # pointwise pairs + SD head (per mel frame)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--seconds", type=float, default=3600.0, help="clip length per segment")
    ap.add_argument("--segments", type=int, default=1, help="segments (independent B=1 calls) per GPU per step")
    ap.add_argument("--batched", action="store_true", help="process the segments of a step as ONE [segments, L] call "
                    "(one reference call: the log-mel mean then couples the batch, tal/asr/models.py:52)")
    ap.add_argument("--cpu-seconds", type=float, default=300.0, help="clip length of the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="skip the per-launch HIP-event timing")
    return ap.parse_args()


def build_model(dev):
    from tal_asrd_amd import SDModel, synth
    model = SDModel()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth.fill_state_dict(shapes)
    own = model.state_dict()
    for k, v in sd.items():
        own[k] = torch.from_numpy(v.copy())
    model.load_state_dict(own)
    return model.to(dev), sd


def dense_layer_algorithmic_bytes(frames):
    """Algorithmic HBM bytes of the dense-layer launches of one SD-path step over `frames` mel frames, summed
    and per launch: every launch reads its activations (+ the residual for the second pointwise layer of a
    TDSBlock) and its weights once and writes its output once.  (For comparison with roofline.traffic.)"""
    def f(t):
        return (t - 21) // 2 + 1
    t1 = f(frames); t2 = f(t1); t3 = f(t2)
    total = 0.0
    launches = 0
    for t, c, depth in ((t1, 800, 2), (t2, 1120, 3), (t3, 1440, 6)):
        act = 4.0 * t * c
        w = 4.0 * c * c
        total += depth * ((act + w + act) + (act + w + act + act))      # relu layer; residual layer
        launches += 2 * depth
    total += 4.0 * (t3 * 1440 + 128 * 1440 + t3 * 128)                   # 1440 -> 128 features
    total += 4.0 * (t3 * 128 + 6008 * 128) + 8.0 * t3 * 38                # 128 -> 6008 logits, arg-max partials only
    launches += 2
    return total, launches


def cpu_baseline(sd, seconds):
    """The oracle (CPU restatement of the reference's PyTorch-CPU path) on a bounded sample."""
    from oracle import tal_oracle as O
    from tal_asrd_amd import synth
    L = int(seconds * 16000)
    audio = synth.synth_audio_batch(1, L, 1234)
    frames = 1 + L // 160
    cores = torch.get_num_threads()
    O.sd_path(audio, sd)  # warm-up
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        O.sd_path(audio, sd)
        times.append(time.perf_counter() - t0)
    med = sorted(times)[1]
    return {"value": frames / med, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%.0f s synthetic clip (%d frames), torch-CPU fp32, %d threads, median of 3 after 1 warm-up"
                      % (seconds, frames, cores)}


def main():
    args = parse()
    import __graft_entry__ as g
    g.build()
    from tal_asrd_amd import _native, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # TAL_BENCH_BACKEND=gloo is a plumbing self-test only (N ranks sharing the visible GPUs, results
    # staged through host memory); real runs use RCCL ("nccl") with one GPU per rank.
    backend = os.environ.get("TAL_BENCH_BACKEND", "nccl")
    dev = torch.device("cuda", local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    model, sd = build_model(dev)
    if dist is not None:
        # weights come from rank 0 over RCCL/xGMI (one flat broadcast per tensor, start-up only)
        for t in list(model.parameters()) + list(model.buffers()):
            if backend == "nccl":
                dist.broadcast(t.data, src=0)
            else:
                h = t.data.cpu()
                dist.broadcast(h, src=0)
                t.data.copy_(h)

    L = int(args.seconds * 16000)
    frames = 1 + L // 160
    clips = [torch.from_numpy(synth.synth_audio_batch(1, L, 1234 + rank * args.segments + i)).to(dev)
             for i in range(args.segments)]
    if args.batched:
        clips = [torch.cat(clips, dim=0)]
    torch.cuda.synchronize()
    # PCIe-inclusive figure (reported beside `value`, never as `value`): one pinned-host -> device copy
    h2d_ms = None
    if rank == 0:
        host = clips[0].cpu().pin_memory()
        torch.cuda.synchronize()
        t_h = time.perf_counter()
        _tmp = host.to(dev, non_blocking=True)
        torch.cuda.synchronize()
        h2d_ms = 1e3 * (time.perf_counter() - t_h)
        del _tmp, host

    gather_feat = gather_ids = None
    pending = []

    def step():
        outs = []
        for clip in clips:
            feat, ids = model.speaker_ids(clip)
            outs.append((feat, ids))
        if dist is not None:
            feat = torch.cat([o[0].reshape(-1, o[0].shape[-1]) for o in outs])
            ids = torch.cat([o[1].reshape(-1) for o in outs])
            if backend != "nccl":
                feat, ids = feat.cpu(), ids.cpu()
            # Result gather to rank 0 is asynchronous (RCCL's own stream): it overlaps the next step's
            # compute instead of serialising 23 MB x (N-1) of xGMI traffic behind every step; all
            # pending gathers are waited for inside the timed region.  `keep` pins the source tensors.
            pending.append((dist.gather(feat, gather_feat if rank == 0 else None, dst=0, async_op=True),
                            dist.gather(ids, gather_ids if rank == 0 else None, dst=0, async_op=True), feat, ids))
        return outs

    def drain():
        while pending:
            w1, w2, _, _ = pending.pop(0)
            w1.wait()
            w2.wait()

    with torch.no_grad():
        if dist is not None and rank == 0:
            f0, i0 = model.speaker_ids(clips[0])
            n_rows = f0.shape[-2] * args.segments
            gdev = dev if backend == "nccl" else torch.device("cpu")
            gather_feat = [torch.empty(n_rows, f0.shape[-1], device=gdev) for _ in range(world)]
            gather_ids = [torch.empty(n_rows, dtype=torch.int32, device=gdev) for _ in range(world)]
        for _ in range(args.warmup):
            step()
        drain()
        lib = _native.lib()
        prof = not args.no_prof
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        if prof:
            lib.tal_prof_reset()
            lib.tal_prof_enable(1)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        drain()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if prof:
            lib.tal_prof_enable(0)

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_frames = world * args.segments * frames * args.steps
        line = {
            "metric": "audio frames/sec (16 kHz, 10 ms hop) end-to-end log-mel -> TDS encoder -> diarization head",
            "value": total_frames / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if os.environ.get("TAL_TDS_F32") else "f32 (dense layers: 3 x f16 MFMA hi/lo split, f32 accumulate)",
            "data": "synthetic",
            "config": {"workload": (("%d x %.0f s 16 kHz clips per GPU as one batched call " if args.batched else
                                     "%d x %.0f s 16 kHz clip per GPU, each a whole-episode B=1 call ") % (args.segments, args.seconds)) +
                                   "(BASELINE.json configs[2]; SDModel path of tal/baseline/reconcile.py:76-85: "
                                   "log-mel -> TDS 80-800-1120-1440 -> 128-d feat + argmax over 6008 speakers)",
                       "frames_per_gpu_per_step": args.segments * frames, "weights": "synthetic deterministic",
                       "audio_resident_in_hbm": True},
        }
        if prof:
            ms, n, work = C.c_double(), C.c_int64(), C.c_double()
            kern = {}
            for cls, name in ((0, "gemm_nt_f32"), (1, "gconv_res"), (2, "gconv_s2"), (3, "logmel"), (4, "other")):
                _native.check(lib.tal_prof_collect(cls, C.byref(ms), C.byref(n), C.byref(work)))
                kern[name] = {"ms_total": ms.value, "launches": n.value, "work": work.value}
            gm = kern["gemm_nt_f32"]
            achieved = gm["work"] / (gm["ms_total"] * 1e-3) / 1e12 if gm["ms_total"] > 0 else 0.0
            traffic = None
            try:   # HBM-side bytes per launch from the separate rocprofv3 --pmc passes (profiles/)
                with open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")) as f:
                    traffic = json.load(f)["hbm_bytes_per_launch"]
            except Exception:
                pass
            f32_only = bool(os.environ.get("TAL_TDS_F32"))
            peak = FP32_MATRIX_PEAK_TFLOPS if f32_only else F16_MATRIX_PEAK_TFLOPS
            # MFMA flops actually issued: the pointwise layers run as 3 fp16 MFMAs per fp32 product (hi*hi, hi*lo, lo*hi)
            pw = 2.0 * POINTWISE_MAC_PER_FRAME * frames * args.segments * args.steps
            issued = gm["work"] + (0.0 if f32_only else 2.0 * pw)
            line["roofline"] = {"bound": "mfma",
                                "kernel": "tal::gemm_glds_kernel (dense layers; TDS pointwise layers in the fp16x3 form: fp32 "
                                          "products as 3 f16 MFMAs, fp32 accumulate)" if not f32_only else
                                          "tal::gemm_glds_kernel (fp32 MFMA dense layer, all epilogues)",
                                "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                                "frac": achieved / peak, "traffic": traffic,
                                "achieved_is": "algorithmic fp32 flops (2 M N K per dense layer) / HIP-event time",
                                "issued_mfma_tflops": issued / (gm["ms_total"] * 1e-3) / 1e12 if gm["ms_total"] > 0 else 0.0,
                                "issued_frac": (issued / (gm["ms_total"] * 1e-3) / 1e12 / peak) if gm["ms_total"] > 0 else 0.0,
                                "fp32_matrix_peak": FP32_MATRIX_PEAK_TFLOPS,
                                "avg_launch_ms": gm["ms_total"] / max(gm["launches"], 1),
                                "launches": gm["launches"],
                                "algorithmic_flops_per_launch": gm["work"] / max(gm["launches"], 1),
                                "algorithmic_bytes_per_launch": (lambda tb: tb[0] / tb[1])(dense_layer_algorithmic_bytes(frames))}
            tot = sum(k["ms_total"] for k in kern.values())
            line["kernel_time_share"] = {k: (v["ms_total"] / tot if tot else 0.0) for k, v in kern.items()}
            line["kernel_ms_per_step"] = {k: v["ms_total"] / args.steps for k, v in kern.items()}
        if h2d_ms is not None:
            line["h2d_ms_per_clip"] = h2d_ms
            line["value_including_h2d"] = total_frames / (elapsed + 1e-3 * h2d_ms * args.segments * args.steps)
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sd, args.cpu_seconds)
            line["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
