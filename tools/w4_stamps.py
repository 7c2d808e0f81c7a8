#!/usr/bin/env python3
"""Per-workgroup cycle stamps of the F(4x4) Winograd kernel (diagnostic builds libbmc_hip_w4stampN.so: make -C bmcnet-esr_amd/csrc
libbmc_hip_w4stamp0.so): where a launch's time goes -- start skew between workgroups, prologue, every tile's chunk loop and
epilogue.   BMC_HIP_LIB=.../libbmc_hip_w4stamp0.so python tools/w4_stamps.py [B H W]"""
import ctypes as C
import os
import sys

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
for _ in range(20):
    fn()
torch.cuda.synchronize()
rd = lib._lib.bmc_w4_read_stamps
rd.argtypes = [C.c_void_p]
host = np.zeros((1024, 16), dtype=np.uint64)
assert rd(host.ctypes.data) == 0
NWG = int(os.environ.get('BMC_W4_GRID', 256))
s = host[:NWG].astype(np.int64)
t0 = s[:, 0].min()
ntile = ((s[:, 2:13] > 0).sum(1)) // 2
print("workgroups %d; tiles" % NWG if False else "workgroups; tiles per workgroup: %s" % np.bincount(ntile))
print("start skew (cycles after the first workgroup's start): median %d, p90 %d, max %d" % tuple(np.percentile(s[:, 0] - t0, [50, 90, 100])))
print("prologue: median %d cycles" % np.median(s[:, 1] - s[:, 0]))
for i in range(int(ntile.max())):
    m = ntile > i
    prev = np.where(i == 0, s[:, 1], s[:, 1 + 2 * i])
    loop = s[m, 2 + 2 * i] - prev[m]
    epi = s[m, 3 + 2 * i] - s[m, 2 + 2 * i]
    print("tile %d (%3d workgroups): chunk loop median %6d p90 %6d cycles (ideal 8 x 9216 = 73728), epilogue median %5d" % (i, m.sum(), np.median(loop), np.percentile(loop, 90), np.median(epi)))
end = s[:, 13] - t0
print("end: median %d, max %d cycles after the first start; realtime span of the launch %.1f us" % (np.median(end), end.max(), (s[:, 15].max() - s[:, 14].min()) / 100.0))
clk = (s[:, 13] - s[:, 0]) / np.maximum(1, (s[:, 15] - s[:, 14])) * 100e6
print("in-kernel clock: median %.3f GHz" % (np.median(clk) / 1e9))

# per-wave timeline of workgroup 8: cycles relative to wave 0's stamp 0 of the chunk
rw = lib._lib.bmc_w4_read_wstamps
rw.argtypes = [C.c_void_p]
ws = np.zeros((8, 24, 8), dtype=np.uint64)
assert rw(ws.ctypes.data) == 0
ws = ws.astype(np.int64)
print("per-wave stamps of one workgroup (cycles since the chunk's first stamp of any wave): pair0 | pair1 (after the burst) | pair7 | pair12 | barrier arrival | barrier passed")
for c in (2, 3, 9, 10, 17):
    base = ws[:, c, 0].min()
    print("chunk %2d (length %6d):" % (c, ws[:, c + 1, 0].min() - base))
    for wv in range(8):
        print("   wave %d: %s" % (wv, " ".join("%6d" % (ws[wv, c, k] - base) for k in range(6))))
