#!/usr/bin/env python3
"""The clock the chip holds under the F(4x4) convolution kernel, and the kernel's length in CYCLES: clock-only diagnostic builds
(make -C bmcnet-esr_amd/csrc libbmc_hip_w4ilclk1.so: two stamps around the whole kernel, nothing inside), >= 2 s of back-to-back
launches first (MI355X_MICROARCH.md, DVFS give-back item 6).  Separates "fewer cycles" from "faster": a power-limited chip gives
part of a cycle saving back as a lower clock.   BMC_HIP_LIB=.../libbmc_hip_w4ilclk1.so python tools/w4_clock.py [B H W]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bmcnet-esr_amd"))
import numpy as np
import torch
from bmc_hip import lib
from bmc_hip.ops import ConvSpec, _packed_weight, _src, conv_raw, coutpad

dev = torch.device("cuda:0")
B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8, 180, 240)
Cn = 128
spec, cp = ConvSpec.dense(Cn), coutpad(Cn)
x = torch.randn(B, H, W, Cn, device=dev)
w = torch.randn(1, Cn, Cn, 9, device=dev) * 0.03
bias = torch.randn(1, Cn, device=dev)
wp = _packed_weight(w, spec, None, wino=4)
out = torch.empty(B, H, W, Cn, device=dev)
fn = lambda: conv_raw([_src(x, 0, Cn, 0, None, 0, B)], wp, spec.kpad * 9 * cp, bias, Cn, out.data_ptr(), H * W * Cn, Cn, B, H, W, Cn, 9, relu=True, bpg=B, wino=4)
t0 = time.time()
n = 0
while time.time() - t0 < float(os.environ.get("W4_CLOCK_SECONDS", 2.5)):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    n += 200
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    fn()
e1.record()
torch.cuda.synchronize()
rd = lib._lib.bmc_w4_read_stamps
rd.argtypes = [C.c_void_p]
host = np.zeros((1024, 16), dtype=np.uint64)
assert rd(host.ctypes.data) == 0
s = host[:256].astype(np.int64)
cyc = s[:, 13] - s[:, 0]
wall = (s[:, 15] - s[:, 14]) / 100.0
clk = cyc / np.maximum(wall, 1e-9) / 1e3
print("%d warm launches; launch %.4f ms (events, 200 launches); per workgroup: cycles median %d max %d, wall median %.1f max %.1f us, in-kernel clock median %.3f GHz (p10 %.3f, p90 %.3f)" % (
    n, e0.elapsed_time(e1) / 200, np.median(cyc), cyc.max(), np.median(wall), wall.max(), np.median(clk), np.percentile(clk, 10), np.percentile(clk, 90)))
