#!/usr/bin/env python3
"""Timing of the bf16-plane conv kernel at the C2 shape for an experiment build selected with BMC_HIP_LIB."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bmcnet-esr_amd"))
import torch
from bmc_hip import ops
from bmc_hip.ops import ConvSpec, View
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, H, W, Cn = 8, 180, 240, 128
taps = int(os.environ.get("KB_TAPS", 9)); k = 3 if taps == 9 else 1
x = torch.randn(B, H, W, Cn, device=dev)
w = torch.randn(Cn, Cn, k, k, device=dev) * 0.03; b = torch.zeros(Cn, device=dev)
if os.environ.get("KB_ZERO") == "x":        # same binary, same instruction stream, all-zero activations: what the clock alone is worth
    x.zero_()
elif os.environ.get("KB_ZERO") == "xw":
    x.zero_(); w.zero_()
spec = ConvSpec.dense(Cn)
for mode in sys.argv[1:] or ["bf16x6", "bf16"]:
    iters = int(os.environ.get("KB_ITERS", 30))
    ops.set_math(mode)
    with torch.no_grad():
        for _ in range(5):
            ops.conv([View(x)], w, b, spec, relu=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.conv([View(x)], w, b, spec, relu=True)
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print("%s %s zero=%s taps=%d: %.3f ms  %.1f TFLOP/s" % (os.path.basename(os.environ.get("BMC_HIP_LIB", "libbmc_hip.so")), mode, os.environ.get("KB_ZERO", "-"), taps, ms, 2.0 * B * H * W * Cn * Cn * taps / ms / 1e9), flush=True)
