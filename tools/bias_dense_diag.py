#!/usr/bin/env python3
"""Which 3x3 biases of the bench's model are still within DENSE_FLOOR of zero after a few optimizer steps (the exact-zero rule
of ops.wino_ok keeps the forward launches of those layers on F(2x2))?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "bmcnet-esr_amd")):
    sys.path.insert(0, p)
import torch
import bench
from bmc_hip import ops

dev = torch.device("cuda:0")
wl = bench.Workload(dev, int(os.environ.get("BD_B", 4)), 180, 240, 9, 128, 5, "fp32")
for step in range(int(os.environ.get("BD_STEPS", 3))):
    wl.step()
    torch.cuda.synchronize()
    rows = []
    for n, p in wl.model.named_parameters():
        if n.endswith("bias") and p.dim() == 1:
            a = p.detach().abs()
            rows.append((n, float(a.min()), int((a == 0).sum()), int((a < ops.DENSE_FLOOR).sum()), float(p.grad.abs().min()) if p.grad is not None else -1.0,
                         int((p.grad == 0).sum()) if p.grad is not None else -1))
    bad = [r for r in rows if r[3] > 0]
    print("after step %d: %d of %d bias vectors have elements within %.0e of zero" % (step + 1, len(bad), len(rows), ops.DENSE_FLOOR))
    for r in bad:
        print("    %-50s min|b| %.2e  exact zeros %d  below floor %d | grad: min|g| %.2e exact-zero grads %d" % r)
