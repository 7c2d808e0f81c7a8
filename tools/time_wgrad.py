#!/usr/bin/env python3
"""Isolated timing of the 3x3 128->128 weight gradient: Winograd kernels (csrc/wino4_wgrad.hip F(4x4), csrc/wino_wgrad.hip F(2x2)) vs the pixel-reduction GEMM.
TW_B / TW_H / TW_W select the shape (default: the 4B = 16-sample launches of the C2 step)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "bmcnet-esr_amd")):
    sys.path.insert(0, p)
import torch
from bmc_hip import ops

dev = torch.device("cuda:0")
B, H, W = int(os.environ.get("TW_B", 16)), int(os.environ.get("TW_H", 180)), int(os.environ.get("TW_W", 240))
x = torch.randn(B, H, W, 128, device=dev)
g = torch.randn(B, H, W, 128, device=dev)
spec = ops.ConvSpec.dense(128)
w = torch.zeros(128, 128, 3, 3, device=dev)
b = torch.zeros(128, device=dev)
flops = 2.0 * B * H * W * 128 * 128 * 9


def run(n=20):
    for _ in range(3):
        ops._wgrad_plain(g, x, spec, w, b, 9)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops._wgrad_plain(g, x, spec, w, b, 9)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for mode in ((os.environ["TW_ONLY"],) if os.environ.get("TW_ONLY") else ("winograd4", "winograd", "direct")):
    ops.WINO_WGRAD = mode != "direct"
    ops.WINO4_WGRAD = mode == "winograd4"
    ms = run()
    ops.PROFILE = []
    run(10)
    torch.cuda.synchronize()
    ks = [e0.elapsed_time(e1) for _, _, e0, e1, _ in ops.PROFILE]
    ops.PROFILE = None
    print("   main kernel alone: %.3f ms" % (sum(ks) / len(ks)))
    print("%-9s B%d %dx%d: %.3f ms per weight gradient (kernel + reduction) = %.1f algorithmic TFLOP/s" % (mode, B, H, W, ms, flops / ms / 1e9))
