#!/usr/bin/env python3
"""One-rank RCCL self-test on the GPU box (the multi-GPU bench is the driver's to run): the process group bench.py
creates (backend "nccl" = RCCL, device_id given), the collectives it uses (barrier, all_gather_into_tensor of fp32 rows
and of the fp64 per-rank times) and aaclip_hip.shard's gather on a world of one."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
from aaclip_hip import shard
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
assert dist.get_world_size() == 1 and dist.get_backend() == "nccl"
x = torch.arange(64 * 768, dtype=torch.float32, device=dev).view(64, 768)
out = shard._all_gather(x, 1)
assert torch.equal(out, x)
t = torch.tensor([1.25], device=dev, dtype=torch.float64)
allt = torch.empty(1, device=dev, dtype=torch.float64)
dist.all_gather_into_tensor(allt, t)
assert float(allt[0]) == 1.25
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(shard.gather_rows(x), x)
dist.destroy_process_group()
print("rccl self-test ok: nccl backend, all_gather_into_tensor fp32/fp64, barrier")
