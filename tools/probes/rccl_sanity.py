"""One-rank RCCL sanity check (the multi-GPU bench path needs init_process_group('nccl') + all_gather_into_tensor)."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.arange(12, dtype=torch.float64, device="cuda").reshape(3, 4)
out = torch.empty_like(x)
dist.all_gather_into_tensor(out, x)
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(out, x) and float(t) == 1.5
print("rccl one-rank all_gather_into_tensor / all_reduce / barrier ok")
dist.destroy_process_group()
