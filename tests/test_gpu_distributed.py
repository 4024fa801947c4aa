"""N > 1 data path on the GPU: two ranks share the one visible GPU (gloo rendezvous on 127.0.0.1) and execute ONE
reference call over a [4, L] batch sharded by segment (tal_asrd_amd.distributed: shard_indices, the (sum, count)
all-reduce of the call's log-mel mean, gather_segments); rank 0 checks ids identical and features equal to the
single-process call."""
import os
import socket
import subprocess
import sys

import pytest

from tests.conftest import ROOT, has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_one_reference_call_sharded_by_segment():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_dist_gpu_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "rank 0 ok" in p.stdout and "rank 1 ok" in p.stdout


def test_bench_segments_two_ranks_on_the_real_kernels():
    """The exact code path the driver's multi-GPU run takes -- `bench.py --gpus 2 --workload segments` launched through
    torch.distributed.run -- on the real HIP kernels, with the gloo backend standing in for RCCL because the test box has
    one GPU (both ranks share it; results are staged through host memory): shard_indices, flat weight broadcast, the scalar
    all-reduce of the log-mel mean, ONE packed gather per step, the same-workload one-GPU reference pass."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", TAL_BENCH_BACKEND="gloo")
    env.pop("TAL_BENCH_FAKE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "segments",
           "--segments", "6", "--seconds", "20", "--steps", "2", "--warmup", "1"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["data"] == "synthetic"
    assert line["gathered_segments"] == 6 and line["value"] > 0
    ref = line["one_gpu_same_workload"]
    assert ref["n_gpus"] == 1 and ref["value"] > 0 and line["speedup_vs_one_gpu"] > 0
    assert "roofline" in line and line["roofline"]["traffic_source"]
