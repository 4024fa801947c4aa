"""N>1 path on CPU: world_size 2, gloo backend, launched exactly like the driver launches
bench.py (torch.distributed.run, 127.0.0.1 rendezvous)."""
import os
import subprocess
import sys

from tests.conftest import ROOT


def test_two_rank_sharding_broadcast_gather():
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "tests", "_dist_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "rank 0 ok" in p.stdout and "rank 1 ok" in p.stdout
