#!/usr/bin/env python3
"""World-size-1 RCCL smoke test: process-group creation the way xgpr_amd.dist does it for N > 1,
a float64 sum all-reduce of the sizes the path uses, and a barrier."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for n in (1, 3, 8192, 512 * 8192):
    t = torch.arange(n, dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    assert torch.equal(t, torch.arange(n, dtype=torch.float64, device="cuda"))
e = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(e, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("rccl smoke ok")
