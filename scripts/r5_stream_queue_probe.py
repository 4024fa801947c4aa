"""Which HIP streams share a hardware queue?  Two chains of small dependent kernels on two streams take the time of one chain when the
streams sit on different hardware queues and about twice that when they share one.  python scripts/r5_stream_queue_probe.py [n_streams]"""
import sys, time
import torch
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
K = 40

def chain(i):
    with torch.cuda.stream(streams[i]):
        for _ in range(K):
            torch.cuda._sleep(100000)        # a spin kernel of ~50 us: the chains are GPU-bound, not launch-bound

def timed(idx):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in idx:
        chain(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3

for i in range(n): timed([i])
one = min(timed([0]) for _ in range(3))
print("one chain of %d kernels: %.2f ms; stream ids: %s" % (K, one, [s.stream_id for s in streams]))
print("pairs (ms; ~%.1f = separate queues, ~%.1f = one queue):" % (one, 2 * one))
for i in range(n):
    print("  %2d: " % i + " ".join("%5.2f" % (min(timed([i, j]) for _ in range(2)) if j != i else 0.0) for j in range(n)))
