#!/usr/bin/env python3
"""Accuracy (vs float64) and speed of the three convolution arithmetic modes (fp32 MFMA, bf16, bf16x6 split) on
seeded random operands.  usage: python tools/math_modes.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bmcnet-esr_amd"))
import torch
import torch.nn.functional as F
from bmc_hip import ops
from bmc_hip.ops import ConvSpec, View

dev = torch.device("cuda:0")
torch.manual_seed(0)


def rel(a, b):
    return ((a.double() - b).norm() / b.norm()).item()


def bench(fn, iters=20):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def case(name, B, H, W, cins, Cout, k, big):
    xs = [torch.randn(B, H, W, c, device=dev) for c in cins]
    cin = sum(cins)
    w = torch.randn(Cout, cin, k, k, device=dev) * (1.0 / (cin * k * k) ** 0.5)
    b = torch.randn(Cout, device=dev) * 0.1
    spec = ConvSpec.dense(*cins)
    xcat = torch.cat(xs, -1).permute(0, 3, 1, 2)
    ref = None
    if not big:
        ref = F.conv2d(xcat.double(), w.double(), b.double(), padding=k // 2).permute(0, 2, 3, 1)
        xb = xcat.bfloat16().double()
        refb = F.conv2d(xb, w.bfloat16().double(), b.double(), padding=k // 2).permute(0, 2, 3, 1)
    flops = 2.0 * B * H * W * Cout * cin * k * k
    outs = {}
    for mode in ("fp32", "bf16x6", "bf16"):
        ops.set_math(mode)
        with torch.no_grad():
            y = ops.conv([View(t) for t in xs], w, b, spec)
            ms = bench(lambda: ops.conv([View(t) for t in xs], w, b, spec))
        outs[mode] = y
        msg = "%-26s %-7s %8.3f ms %8.1f TFLOP/s" % (name, mode, ms, flops / ms / 1e9)
        if ref is not None:
            msg += "  rel-L2 vs f64 %.3e" % rel(y, ref)
            if mode == "bf16":
                msg += "  vs f64(bf16 operands) %.3e" % rel(y, refb)
        else:
            msg += "  rel-L2 vs fp32 kernel %.3e" % rel(y, outs["fp32"].double())
        print(msg, flush=True)
    ops.set_math("fp32")


case("3x3 128->128 45x80 b2", 2, 45, 80, [128], 128, 3, False)
case("3x3 144->128 37x53 b3", 3, 37, 53, [16, 128], 128, 3, False)
case("1x1 256->128 45x80 b2", 2, 45, 80, [128, 128], 128, 1, False)
case("3x3 128->32 45x80 b2", 2, 45, 80, [128], 32, 3, False)
case("3x3 128->128 180x240 b8", 8, 180, 240, [128], 128, 3, True)
case("1x1 256->128 180x240 b8", 8, 180, 240, [128, 128], 128, 1, True)
case("1x1 128->128 180x240 b8", 8, 180, 240, [128], 128, 1, True)


def wgrad_case(name, B, H, W, cins, Cout, k, big):
    xs = [torch.randn(B, H, W, c, device=dev) for c in cins]
    cin = sum(cins)
    w = (torch.randn(Cout, cin, k, k, device=dev) * (1.0 / (cin * k * k) ** 0.5)).requires_grad_()
    b = (torch.randn(Cout, device=dev) * 0.1).requires_grad_()
    g = torch.randn(B, H, W, Cout, device=dev)
    spec = ConvSpec.dense(*cins)
    ref = None
    if not big:
        xd = torch.cat(xs, -1).permute(0, 3, 1, 2).double()
        wd, bd = w.detach().double().requires_grad_(), b.detach().double().requires_grad_()
        F.conv2d(xd, wd, bd, padding=k // 2).backward(g.permute(0, 3, 1, 2).double())
        ref = (wd.grad, bd.grad)
    flops = 2.0 * B * H * W * Cout * cin * k * k
    base = None
    for mode in ("fp32", "bf16x6", "bf16"):
        ops.set_math(mode)
        w.grad = b.grad = None
        y = ops.conv([View(t) for t in xs], w, b, spec)
        y.backward(g)
        dw, db = w.grad.clone(), b.grad.clone()
        gs = ops._src(g, 0, Cout, 0, None, 0, B)
        srcs = [ops._src(t, 0, t.shape[-1], 0, None, 0, B) for t in xs]
        ms = bench(lambda: ops.pgemm_raw(gs, srcs, B, H, W, k * k, B, Cout, cin, dev, want_bias=True))
        msg = "wgrad %-22s %-7s %8.3f ms %8.1f TFLOP/s" % (name, mode, ms, flops / ms / 1e9)
        if ref is not None:
            msg += "  dw rel-L2 vs f64 %.3e  db %.3e" % (rel(dw, ref[0]), rel(db, ref[1]))
        else:
            if base is None:
                base = (dw.double(), db.double())
            msg += "  dw rel-L2 vs fp32 kernel %.3e  db %.3e" % (rel(dw, base[0]), rel(db, base[1]))
        print(msg, flush=True)
    ops.set_math("fp32")


wgrad_case("3x3 128->128 45x80 b2", 2, 45, 80, [128], 128, 3, False)
wgrad_case("3x3 144->128 37x53 b3", 3, 37, 53, [16, 128], 128, 3, False)
wgrad_case("1x1 256->128 45x80 b2", 2, 45, 80, [128, 128], 128, 1, False)
wgrad_case("3x3 128->32 45x80 b2", 2, 45, 80, [128], 32, 3, False)
wgrad_case("3x3 128->128 180x240 b8", 8, 180, 240, [128], 128, 3, True)
wgrad_case("1x1 256->128 180x240 b8", 8, 180, 240, [128, 128], 128, 1, True)
