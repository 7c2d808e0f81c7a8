#!/usr/bin/env python3
"""Back-to-back launches of the four MFMA kernels at the C2 shapes in one arithmetic mode (for rocprofv3 --pmc runs).
usage: python tools/kb_modes.py <fp32|bf16x6|bf16> [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bmcnet-esr_amd"))
import torch
from bmc_hip import ops
from bmc_hip.ops import ConvSpec, View
dev = torch.device("cuda:0")
torch.manual_seed(0)
ops.set_math(sys.argv[1])
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
B, H, W, Cn = 8, 180, 240, 128
x = torch.randn(B, H, W, Cn, device=dev); x2 = torch.randn(B, H, W, Cn, device=dev); g = torch.randn(B, H, W, Cn, device=dev)
w3 = torch.randn(Cn, Cn, 3, 3, device=dev) * 0.03; w1 = torch.randn(Cn, Cn, 1, 1, device=dev) * 0.1
w12 = torch.randn(Cn, 2 * Cn, 1, 1, device=dev) * 0.1; b = torch.zeros(Cn, device=dev)
s1, s2 = ConvSpec.dense(Cn), ConvSpec.dense(Cn, Cn)
gs = ops._src(g, 0, Cn, 0, None, 0, B); xs = ops._src(x, 0, Cn, 0, None, 0, B); xs2 = ops._src(x2, 0, Cn, 0, None, 0, B)
def timeit(name, fn, flops):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print("%-8s %-22s %8.3f ms %8.1f TFLOP/s" % (sys.argv[1], name, ms, flops / ms / 1e9), flush=True)
npx = B * H * W
with torch.no_grad():
    timeit("conv3x3 128->128", lambda: ops.conv([View(x)], w3, b, s1, relu=True), 2.0 * npx * Cn * Cn * 9)
    timeit("conv1x1 128->128", lambda: ops.conv([View(x)], w1, b, s1), 2.0 * npx * Cn * Cn)
    timeit("conv1x1 256->128", lambda: ops.conv([View(x), View(x2)], w12, b, s2), 2.0 * npx * Cn * Cn * 2)
    timeit("wgrad3x3 128x128", lambda: ops.pgemm_raw(gs, [xs], B, H, W, 9, B, Cn, Cn, dev, want_bias=True), 2.0 * npx * Cn * Cn * 9)
    timeit("wgrad1x1 128x128", lambda: ops.pgemm_raw(gs, [xs], B, H, W, 1, B, Cn, Cn, dev, want_bias=True), 2.0 * npx * Cn * Cn)
    timeit("wgrad1x1 128x256", lambda: ops.pgemm_raw(gs, [xs, xs2], B, H, W, 1, B, Cn, 2 * Cn, dev, want_bias=True), 2.0 * npx * Cn * Cn * 2)
