#!/usr/bin/env python3
"""Is a workload slower when it is not the first one of its process / when the reducer is in the step?  (diagnostic for the
`dist1` leg of bench.py)"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "bmcnet-esr_amd"))
import torch
import torch.distributed as dist
import bench
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29547")
order = sys.argv[1] if len(sys.argv) > 1 else "pdpd"
if "d" in order:
    dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
for i, c in enumerate(order):
    for wu in (2, 4):
        r = bench.extra_train(dev, "x", 4, 180, 240, 9, "fp32", 5, wu, use_dist=(c == "d"))
        print(i, "reducer" if c == "d" else "plain  ", "warmup", wu, r["ms_per_step"], flush=True)
if dist.is_initialized():
    dist.destroy_process_group()
