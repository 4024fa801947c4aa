"""HIP streams spread over the device's hardware queues.

A HIP stream is served by one of a few hardware queues (four per process on this stack), and the runtime deals streams to queues in
an order a caller cannot choose or ask for: of ten streams created one after the other on an MI355X, the 4th and 5th, the 3rd and
6th, the 2nd and 7th ... share a queue (profiles/r5_stream_hardware_queues.txt).  Two chains of dependent launches on streams that
share a queue run one after the other, not side by side -- System.transcribe_unaligned_many with four decode chains took 3.3 s or
5.0 s for the same corpus depending on which streams its threads happened to get.

`spread(device, k)` returns k streams on as many DIFFERENT hardware queues as there are, found by measurement, once per process and
device: two short chains of spin kernels take the time of one when their streams sit on different queues and twice that when they
share one.  Speed only: any set of streams is correct.

The probe costs ~40 short timed looks (a few tens of milliseconds, with device-wide synchronisations) the first time a process asks;
it measures by timing, so other GPU work running during it can mislead it: when its own baseline (one chain, three looks) is not
stable, or the result is implausible, the pool is treated as ONE class and the probe is repeated at the next call instead of being
cached.  Pooled streams are LEASED: `spread` hands a stream to one caller at a time (`release` gives them back); a second caller
that overlaps the first gets the remaining pooled streams and then fresh ones, never a stream somebody else is running chains on.
"""
import threading
import time

import torch

_lock = threading.Lock()
_classes = {}            # device index -> list of queue classes, each a list of torch.cuda.Stream
_leased = set()          # ids of pooled streams handed out by spread() and not yet released
POOL = 12                # streams probed per device
_SPIN = 60000            # spin-kernel length (device clock ticks, ~30 us)
_CHAIN = 10


def _chain(stream):
    with torch.cuda.stream(stream):
        for _ in range(_CHAIN):
            torch.cuda._sleep(_SPIN)


def _timed(dev, streams):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for s in streams:
        _chain(s)
    torch.cuda.synchronize(dev)
    return time.perf_counter() - t0


def classes(device):
    """-> the device's probed streams grouped by hardware queue (a list of lists; one list when the probe cannot tell them apart)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    with _lock:
        if idx in _classes:
            return _classes[idx]
        with torch.cuda.device(idx):
            pool = [torch.cuda.Stream(device=dev) for _ in range(POOL)]
            if not hasattr(torch.cuda, "_sleep"):                 # (no spin kernel to measure with: every stream in one class --
                _classes[idx] = [pool]                            #  spread() then deals the pool in creation order, as before)
                return _classes[idx]
            for s in pool[:2]:
                _timed(dev, [s])                                  # (first launches: code upload, clocks)
            looks = [_timed(dev, [pool[0]]) for _ in range(3)]
            one = min(looks)
            if max(looks) > 1.5 * one:                            # the baseline itself is unstable (other work on the device):
                return [pool]                                     # one class now, NOT cached -- the next call probes again
            out = []
            for s in pool:
                for c in out:
                    # same queue: the two chains take ~2x one chain; different queues: ~1x (the midpoint decides; best of three:
                    # chains that share a queue never overlap, so one overlapping look settles it)
                    if min(_timed(dev, [c[0], s]) for _ in range(3)) > 1.5 * one:
                        c.append(s)
                        break
                else:
                    out.append([s])
            if len(out) > 8:                                      # (implausible: timing noise) -- one class, probed again next time
                return [pool]
        _classes[idx] = out
        return out


def spread(device, k):
    """k streams, dealt round-robin over the hardware queues (the first min(k, queues) of them pairwise on different queues).
    Pooled streams are leased to the caller until `release(streams)`; streams another caller holds are skipped."""
    cl = classes(device)
    order = []
    with _lock:
        free = [[s for s in c if id(s) not in _leased] for c in cl]
        depth = 0
        while len(order) < k:
            took = False
            for c in free:
                if depth < len(c) and len(order) < k:
                    order.append(c[depth])
                    _leased.add(id(c[depth]))
                    took = True
            depth += 1
            if not took:                                          # more streams asked for than the pool has free: fresh ones
                order.append(torch.cuda.Stream(device=torch.device(device)))
    return order


def release(streams):
    """Give leased streams back to the pool (fresh streams that never were in it are ignored)."""
    with _lock:
        for s in streams:
            _leased.discard(id(s))
