import os, sys, torch, torch.distributed as dist
sys.path.insert(0, "/root/repo")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29555")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from tal_asrd_amd import distributed as D
# with world 1 gather_segments short-circuits; exercise the collectives it uses directly
meta = torch.full((4, 2), -1, dtype=torch.int64, device=dev); dist.all_reduce(meta, op=dist.ReduceOp.MAX)
buf = torch.randn(10, 128, device=dev); out = [torch.empty_like(buf)]; dist.gather(buf, out, dst=0); assert torch.equal(out[0], buf)
ib = torch.randint(0, 6008, (10,), dtype=torch.int32, device=dev); out = [torch.empty_like(ib)]; dist.gather(ib, out, dst=0); assert torch.equal(out[0], ib)
st = torch.tensor([1.0, 2.0], dtype=torch.float64, device=dev); print(D.allreduce_logmel_stats(st))
lin = torch.nn.Linear(4, 4).to(dev); D.broadcast_module(lin)
t = torch.tensor([1.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier(); dist.destroy_process_group(); print("nccl world-1 collectives ok")
