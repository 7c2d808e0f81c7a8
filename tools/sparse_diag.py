#!/usr/bin/env python3
"""The sparse-recording case of tests/test_gpu_r5.py (zero biases, 0.026 events per pixel, BMCNet(4,128,2), 3 windows, 180x240)
through the three 3x3 kernel families -- default dispatch, F(2x2) everywhere, direct kernel everywhere -- every parameter
gradient against the FLOAT64 CPU oracle, with the float32 CPU oracle's own distance beside it (the comparison's noise floor)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "bmcnet-esr_amd")):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F
from bmc_hip import ops
from models.BMCNet import BMCNet
from oracle import bmc_oracle as O
from test_gpu_r5 import sparse_frames

scale, n_c, n_b, B, H, W, NW = 4, 128, 2, 1, 180, 240, 3
dev = torch.device("cuda:0")
rel = lambda a, b: float((a.detach().cpu().double() - b.detach().cpu().double()).norm() / b.detach().cpu().double().norm())


def build():
    torch.manual_seed(501)
    m = BMCNet(scale, n_c, n_b)
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(2.0)
    return m


m = build()
g = torch.Generator().manual_seed(503)
frames = sparse_frames(B, NW + 1, H, W, g)
gts = sparse_frames(B, NW + 1, scale * H, scale * W, g, rate=0.06 / 4)
xs = [frames[:, i:i + 2].transpose(1, 2) for i in range(NW)]
torch.set_num_threads(min(16, os.cpu_count() or 1))
grads = {}
for name, dt in (("fp32", torch.float32), ("fp64", torch.float64)):
    seen = {}
    params = {k: seen.setdefault(v.data_ptr(), v.detach().to(dt).clone().requires_grad_()) for k, v in m.state_dict().items()}
    loss, _, _ = O.bptt_loss(params, [x.to(dt) for x in xs], [gts[:, i + 1].to(dt) for i in range(NW)], n_c, scale)
    loss.backward()
    grads[name] = {k: v.grad for k, v in params.items() if v.grad is not None}
short = lambda n: n.replace("neuro.para_reschunk.0.", "")
floor = sorted(((rel(grads["fp32"][n], grads["fp64"][n]), n) for n, _ in m.named_parameters() if n in grads["fp64"]), reverse=True)
print("float32 CPU oracle vs float64: %s" % [(short(n), "%.1e" % e) for e, n in floor[:4]])
for name, w, w4 in (("default dispatch", True, True), ("F(2x2) everywhere", True, False), ("direct kernel everywhere", False, False)):
    ops.WINO, ops.WINO4 = w, w4
    mm = build().to(dev)
    z = lambda c: torch.zeros(B, c, H, W, device=dev)
    st = (z(n_c), z(n_c), z(n_c), z(2 * scale * scale))
    loss = 0
    for i in range(NW):
        st = mm(xs[i].to(dev), *st, i == 0)
        loss = loss + F.mse_loss(st[-1], gts[:, i + 1].to(dev))
    loss.backward()
    torch.cuda.synchronize()
    e64 = sorted(((rel(p.grad, grads["fp64"][n]), n) for n, p in mm.named_parameters() if n in grads["fp64"]), reverse=True)
    e32 = sorted(((rel(p.grad, grads["fp32"][n]), n) for n, p in mm.named_parameters() if n in grads["fp64"]), reverse=True)
    print("%-26s vs float64: %s | vs the float32 oracle: %s" % (name, [(short(n), "%.1e" % e) for e, n in e64[:3]], [(short(n), "%.1e" % e) for e, n in e32[:2]]))
