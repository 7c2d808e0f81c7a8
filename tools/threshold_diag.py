#!/usr/bin/env python3
"""Which side of a threshold is closer to float64?  The two-window step of tests/test_gpu_r5.py at one frame size with the
value-free BIE attention and the side stream forced on / off, HIP gradients and the fp32 CPU oracle's both against the
float64 oracle (the fp32 oracle's distance is the noise floor of the comparison)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "bmcnet-esr_amd")):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F
from bmc_hip import bie, ops
from models.BMCNet import BMCNet
from oracle import bmc_oracle as O

H, W = (int(v) for v in sys.argv[1:3]) if len(sys.argv) >= 3 else (88, 96)
B, n_b, scale, n_c, seed = 2, 1, 4, 128, 520
dev = torch.device("cuda:0")
rel = lambda a, b: float((a.detach().cpu().double() - b.detach().cpu().double()).norm() / b.detach().cpu().double().norm())


def build():
    torch.manual_seed(seed)
    m = BMCNet(scale, n_c, n_b)
    gb = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(2.0)
        for n, p in m.named_parameters():
            if n.endswith("bias") and p.dim() == 1:
                p.add_((torch.rand(p.shape, generator=gb) - 0.5) * 2e-2)
    return m


m = build()
g = torch.Generator().manual_seed(seed + 2)
frames = torch.poisson(torch.full((B, 3, 2, H, W), 0.284), generator=g)
gts = torch.poisson(torch.full((B, 3, 2, scale * H, scale * W), 0.284), generator=g)
xs = [frames[:, i:i + 2].transpose(1, 2) for i in range(2)]
torch.set_num_threads(min(16, os.cpu_count() or 1))
grads = {}
for name, dt in (("fp32", torch.float32), ("fp64", torch.float64)):
    seen = {}
    params = {k: seen.setdefault(v.data_ptr(), v.detach().to(dt).clone().requires_grad_()) for k, v in m.state_dict().items()}
    loss, _, _ = O.bptt_loss(params, [x.to(dt) for x in xs], [gts[:, 1].to(dt), gts[:, 2].to(dt)], n_c, scale)
    loss.backward()
    grads[name] = {k: v.grad for k, v in params.items() if v.grad is not None}
names = [n for n, _ in m.named_parameters() if n in grads["fp64"]]
floor = sorted(((rel(grads["fp32"][n], grads["fp64"][n]), n) for n in names), reverse=True)
print("%dx%d  fp32 oracle vs fp64 oracle: %s" % (H, W, [(n.replace("neuro.para_reschunk.0.", ""), "%.1e" % e) for e, n in floor[:4]]))
for vfree, side in ((False, "0"), (True, "0"), (False, "1"), (True, "1")):
    bie.VFREE, bie.VFREE_MIN_PIXELS, ops.WGRAD_SIDE = vfree, 0, side
    mm = build().to(dev)
    z = lambda c: torch.zeros(B, c, H, W, device=dev)
    st = (z(n_c), z(n_c), z(n_c), z(2 * scale * scale))
    loss = 0
    for i in range(2):
        st = mm(xs[i].to(dev), *st, i == 0)
        loss = loss + F.mse_loss(st[-1], gts[:, i + 1].to(dev))
    loss.backward()
    torch.cuda.synchronize()
    e64 = sorted(((rel(p.grad, grads["fp64"][n]), n) for n, p in mm.named_parameters() if n in grads["fp64"]), reverse=True)
    e32 = sorted(((rel(p.grad, grads["fp32"][n]), n) for n, p in mm.named_parameters() if n in grads["fp64"]), reverse=True)
    print("value-free %-5s side stream %s: vs fp64 %s | vs fp32 oracle worst %.1e" % (
        vfree, side, [(n.replace("neuro.para_reschunk.0.", ""), "%.1e" % e) for e, n in e64[:3]], e32[0][0]))
