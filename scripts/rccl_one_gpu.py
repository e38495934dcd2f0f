"""Experiment: can two RCCL ranks share ONE GPU?  (If yes, the RCCL transport of the sharded time stepper can be exercised
on the single-GPU box.)  Run under torch.distributed.run with 2 processes."""
import os, sys
import torch, torch.distributed as dist
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    r = dist.get_rank()
    t = torch.ones(4, device="cuda") * (r + 1)
    dist.all_reduce(t)
    print("RCCL_ONE_GPU allreduce ok", r, t.tolist(), flush=True)
    if r == 0:
        dist.send(torch.arange(3, device="cuda", dtype=torch.float64), 1)
    else:
        x = torch.zeros(3, device="cuda", dtype=torch.float64); dist.recv(x, 0); print("RCCL_ONE_GPU recv ok", x.tolist(), flush=True)
    dist.destroy_process_group()
except Exception as e:
    print("RCCL_ONE_GPU failed:", repr(e)[:300], flush=True)
    sys.exit(0)
