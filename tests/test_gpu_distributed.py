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
