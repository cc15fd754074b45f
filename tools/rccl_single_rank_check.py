import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dist.barrier()
x = torch.arange(6, dtype=torch.float64, device="cuda").reshape(2, 3)
parts = [torch.empty_like(x)]
dist.all_gather(parts, x)
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("rccl ok", parts[0].sum().item(), t.item())
dist.destroy_process_group()
