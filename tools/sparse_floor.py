#!/usr/bin/env python3
"""Noise floor of the sparse-recording parity test (tests/test_gpu_r5.py): the CPU oracle in float32 against the same oracle in
float64 on the test's own inputs (zero biases, 0.026 events per pixel, 3 windows, BMCNet(4,128,2) at 180x240).  CPU only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "bmcnet-esr_amd")):
    sys.path.insert(0, p)
import torch
from oracle import bmc_oracle as O

H, W = (int(v) for v in sys.argv[1:3]) if len(sys.argv) >= 3 else (180, 240)
scale, n_c, n_b, B, NW = 4, 128, 2, 1, 3


def sparse_frames(B, L, H, W, gen, rate=0.06, blobs=5, radius=0.16):
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    out = torch.zeros(B, L, 2, H, W)
    for b in range(B):
        mask = torch.zeros(H, W, dtype=torch.bool)
        for _ in range(blobs):
            cy, cx = torch.rand(2, generator=gen) * torch.tensor([H, W], dtype=torch.float32)
            mask |= (yy - cy) ** 2 + (xx - cx) ** 2 < (radius * min(H, W)) ** 2
        out[b] = torch.poisson(torch.full((L, 2, H, W), rate), generator=gen) * mask
    return out


import importlib.util
spec = importlib.util.spec_from_file_location("ref_models", os.path.join(ROOT, "tests", "golden", "ref_stubs.py")) if False else None
# parameters: the same construction as the test, without the HIP package (plain torch modules are not needed: only shapes + init)
sys.path.insert(0, os.path.join(ROOT, "bmcnet-esr_amd"))
os.environ.setdefault("BMC_HIP_LIB", os.path.join(ROOT, "bmcnet-esr_amd", "csrc", "libbmc_hip.so"))
from models.BMCNet import BMCNet
torch.manual_seed(501)
m = BMCNet(scale, n_c, n_b)
with torch.no_grad():
    for p in m.parameters():
        p.mul_(2.0)
seen = {}
p32 = {k: seen.setdefault(v.data_ptr(), v.detach().clone().requires_grad_()) for k, v in m.state_dict().items()}
seen = {}
p64 = {k: seen.setdefault(v.data_ptr(), v.detach().double().clone().requires_grad_()) for k, v in m.state_dict().items()}
g = torch.Generator().manual_seed(503)
frames = sparse_frames(B, NW + 1, H, W, g)
gts = sparse_frames(B, NW + 1, scale * H, scale * W, g, rate=0.06 / 4)
xs = [frames[:, i:i + 2].transpose(1, 2) for i in range(NW)]
torch.set_num_threads(min(16, os.cpu_count() or 1))
t0 = time.time()
l32, _, _ = O.bptt_loss(p32, xs, [gts[:, i + 1] for i in range(NW)], n_c, scale)
l32.backward()
print("fp32 oracle: %.1f s" % (time.time() - t0))
t0 = time.time()
l64, _, _ = O.bptt_loss(p64, [x.double() for x in xs], [gts[:, i + 1].double() for i in range(NW)], n_c, scale)
l64.backward()
print("fp64 oracle: %.1f s" % (time.time() - t0))
errs = {}
for k in p32:
    if p32[k].grad is not None:
        errs[k] = float((p32[k].grad.double() - p64[k].grad).norm() / p64[k].grad.norm())
worst = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
print("loss fp32 %.8e fp64 %.8e" % (l32.item(), l64.item()))
print("fp32 oracle vs fp64 oracle, worst parameter gradients:", [(k, "%.1e" % e) for k, e in worst])
