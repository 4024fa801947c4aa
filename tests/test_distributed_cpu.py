"""N>1 path on CPU: world_size 2, gloo backend, launched exactly like the driver launches
bench.py (torch.distributed.run, 127.0.0.1 rendezvous)."""
import os
import socket
import subprocess
import sys

import pytest

from tests.conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_sharding_broadcast_gather():
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_dist_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "rank 0 ok" in p.stdout and "rank 1 ok" in p.stdout


def test_bench_entry_self_launches_two_ranks_and_gathers_segments():
    """`python bench.py --gpus 2` typed as is: the script starts its own two rank processes, shards the segments of the
    configs[3] workload with distributed.shard_indices and returns them through distributed.gather_segments.  CPU
    plumbing run (TAL_BENCH_FAKE: the per-segment GPU work is replaced by deterministic tensors; the JSON line says so)."""
    import json
    env = dict(os.environ, TAL_BENCH_FAKE="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--segments", "5",
           "--seconds", "30"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "strong"
    assert line["gathered_segments"] == 5 and "FAKE" in line["data"]
    assert "configs[3]" in line["config"]["workload"]
    assert line["value"] > 0 and abs(line["config"]["frames_per_step"] - 5 * 3001) < 1


def test_bench_entry_more_ranks_than_segments():
    """An empty rank (3 ranks, 2 segments) goes through gather_segments without a sample tensor."""
    import json
    env = dict(os.environ, TAL_BENCH_FAKE="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "0", "--segments", "2",
           "--seconds", "30"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 3 and line["gathered_segments"] == 2


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_falls_back_to_gloo_when_rccl_raises(launcher):
    """A run whose RCCL communicator cannot be built (simulated: TAL_BENCH_RCCL_FAIL raises where init_process_group("nccl")
    would) continues on gloo and says so in its line -- under bench.py's own launcher and under torch.distributed.run."""
    import json
    env = dict(os.environ, TAL_BENCH_FAKE="1", TAL_BENCH_RCCL_FAIL="1", MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TAL_BENCH_BACKEND"):
        env.pop(k, None)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--segments", "4", "--seconds", "30"]
    if launcher == "self":
        cmd = [sys.executable] + tail
    else:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + tail
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["gathered_segments"] == 4
    assert line["collective_backend"].startswith("gloo (RCCL unusable")


def _bench_env(**kw):
    env = dict(os.environ, TAL_BENCH_FAKE="1", MASTER_ADDR="127.0.0.1", **kw)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TAL_BENCH_BACKEND", "TORCHELASTIC_USE_AGENT_STORE"):
        env.pop(k, None)
    return env


def _last_line(p):
    import json
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_eight_rank_plan_of_configs3():
    """The exact plan of `bench.py --gpus 8 --workload segments` on an 8-GPU node -- 64 segments, 8 per rank by
    distributed.shard_indices, the gather plan's one all-reduce, one gather per step, the scalar mean all-reduce, rank 0's
    same-workload reference pass -- as a CPU plumbing run (gloo, TAL_BENCH_FAKE), under the launcher the driver uses."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--seconds", "30"]
    line = _last_line(subprocess.run(cmd, env=_bench_env(OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=600))
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["gathered_segments"] == 64
    assert line["config"]["frames_per_step"] == 64 * 3001 and "configs[3]" in line["config"]["workload"]
    assert line["collective_backend"] == "gloo" and "speedup_vs_one_gpu" in line
    from tal_asrd_amd.distributed import shard_indices
    shares = [shard_indices(64, r, 8, weights=[1.0] * 64) for r in range(8)]
    assert all(len(s) == 8 for s in shares) and sorted(sum(shares, [])) == list(range(64))


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_goes_to_gloo_on_every_rank_when_one_rank_cannot_use_rccl(launcher):
    """RCCL fails on rank 1 ONLY (its probe child exits non-zero): the outcome is agreed through the rendezvous store, so rank 0
    -- whose own probe was fine -- moves to gloo as well instead of waiting in an RCCL group rank 1 never joins."""
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--segments", "4", "--seconds", "30"]
    cmd = [sys.executable] + tail if launcher == "self" else \
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", str(_free_port())] + tail
    line = _last_line(subprocess.run(cmd, env=_bench_env(TAL_BENCH_RCCL_FAIL="rank1"), capture_output=True, text=True, timeout=240))
    assert line["n_gpus"] == 2 and line["gathered_segments"] == 4
    assert line["collective_backend"].startswith("gloo (RCCL unusable: rank 1")


def test_bench_survives_an_rccl_communicator_that_hangs():
    """A communicator that never comes up (probe children that do not return) costs the probe's deadline -- here 3 s, 45 s by
    default -- not torch's 300 s collective watchdog per rank: the children are killed and the run continues on gloo."""
    import time as _time
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--segments", "4",
           "--seconds", "30"]
    t0 = _time.time()
    line = _last_line(subprocess.run(cmd, env=_bench_env(TAL_BENCH_RCCL_FAIL="hang", TAL_BENCH_PROBE_TIMEOUT="3"),
                                     capture_output=True, text=True, timeout=240))
    assert _time.time() - t0 < 120
    assert line["gathered_segments"] == 4 and "no communicator + all-reduce within 3 s" in line["collective_backend"]
