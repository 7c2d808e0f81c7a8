#!/usr/bin/env python3
"""Isolated timing of the 3x3 128->128 convolution launch: F(4x4) (csrc/wino4.hip) vs F(2x2) (csrc/wino.hip) vs direct.
usage: python tools/time_wino4.py [B H W]   (default 8 180 240: the step's dominant launch)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bmcnet-esr_amd"))
import torch
from bmc_hip import ops
from bmc_hip.ops import ConvSpec, _packed_weight, _src, conv_raw, coutpad

dev = torch.device("cuda:0")
B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8, 180, 240)
Cn, ITERS = 128, int(os.environ.get("KB_ITERS", 100))
CIN = int(os.environ.get("KB_CIN", Cn))
spec, cp = ConvSpec.dense(CIN), coutpad(Cn)
torch.manual_seed(0)
x = torch.randn(B, H, W, CIN, device=dev)
res = torch.randn(B, H, W, Cn, device=dev)
w = torch.randn(1, Cn, CIN, 9, device=dev) * 0.03
bias = torch.randn(1, Cn, device=dev)
flops = 2.0 * B * H * W * Cn * 9 * CIN
ref = None
ONLY4 = os.environ.get("W4_ONLY") == "1"          # ablation libraries (BMC_HIP_LIB=...libbmc_hip_w4ablN.so): results are wrong by design
KINDS = [int(v) for v in os.environ.get("KB_KINDS", "0,2,4").split(",")]      # KB_KINDS=0,2: small launches the F(4x4) kernel does not take
for name, wino in ((("F(4x4)", 4),) if ONLY4 else tuple(kv for kv in (("direct", 0), ("F(2x2)", 2), ("F(4x4)", 4)) if kv[1] in KINDS)):
    wp = _packed_weight(w, spec, None, wino=wino)
    out = torch.empty(B, H, W, Cn, device=dev)
    for use_res in (False, True):
        fn = lambda: conv_raw([_src(x, 0, CIN, 0, None, 0, B)], wp, spec.kpad * 9 * cp, bias, Cn, out.data_ptr(), H * W * Cn, Cn, B, H, W, Cn, 9,
                              relu=not use_res, residual=_src(res, 0, Cn, 0, None, 0, B) if use_res else None, bpg=B, wino=wino)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(ITERS):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / ITERS
        mult = {0: 1.0, 2: 16 / 36, 4: 36 / 144}[wino]
        print("%-8s %s  %8.4f ms  %7.1f algorithmic TFLOP/s  executed %6.1f TFLOP/s = %.3f of the fp32 MFMA peak" % (
            name, "residual" if use_res else "relu    ", ms, flops / ms / 1e9, flops * mult / ms / 1e9, flops * mult / ms / 1e9 / 157.3))
    if ONLY4:
        continue
    if ref is None:
        ref = out.clone()
    else:
        print("         rel-L2 vs direct: %.2e" % float((out - ref).norm() / ref.norm()))
