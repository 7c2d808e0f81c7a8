#!/usr/bin/env python3
"""Diagnostic (libbmc_hip_diag.so): per-workgroup cycle/wall stamps of the conv kernel -> in-kernel clock,
block duration distribution, concurrency. Not part of the product path."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bmcnet-esr_amd"))
import torch
import bmc_hip.lib as L
diag = ctypes.CDLL(os.path.join(ROOT, "bmcnet-esr_amd", "csrc", "libbmc_hip_diag%s.so" % os.environ.get("BMC_DIAG_MODE", "")))
# route the binding to the diag library
for name in L.EXPORTS:
    fn = getattr(diag, name); old = getattr(L._lib, name)
    fn.argtypes, fn.restype = old.argtypes, old.restype
L._conv = diag.bmc_conv; L._pack_w = diag.bmc_pack_weight
from bmc_hip import ops
from bmc_hip.ops import ConvSpec, View
import numpy as np
dev = torch.device("cuda:0")
B, H, W, Cn = int(os.environ.get("KB_B", 8)), 180, 240, 128
taps = int(os.environ.get("KB_TAPS", 9))
k = 3 if taps == 9 else 1
x = torch.randn(B, H, W, Cn, device=dev)
w = torch.randn(Cn, Cn, k, k, device=dev) * 0.03; b = torch.zeros(Cn, device=dev)
spec = ConvSpec.dense(Cn)
nblk = min(B * ((H + 7) // 8) * ((W + 15) // 16), 768)
buf = torch.zeros(nblk * 4, dtype=torch.int64, device=dev)
diag.bmc_diag_set_buffer.argtypes = [ctypes.c_void_p]
with torch.no_grad():
    for _ in range(20):
        ops.conv([View(x)], w, b, spec, relu=True)
    torch.cuda.synchronize()
    assert diag.bmc_diag_set_buffer(buf.data_ptr()) == 0
    ops.conv([View(x)], w, b, spec, relu=True)
    torch.cuda.synchronize()
d = buf.cpu().numpy().reshape(nblk, 4).astype(np.float64)
cyc = d[:, 1] - d[:, 0]; real = (d[:, 3] - d[:, 2]) / 100e6
clk = cyc / real / 1e9
t0 = d[:, 2].min()
print("blocks", nblk, "kernel span ms", (d[:, 3].max() - t0) / 100e6 * 1e3)
print("in-kernel clock GHz: median %.3f  p10 %.3f p90 %.3f" % (np.median(clk), np.percentile(clk, 10), np.percentile(clk, 90)))
print("block duration us: median %.1f p10 %.1f p90 %.1f max %.1f" % tuple(np.percentile(real * 1e6, [50, 10, 90, 100])))
print("block cycles: median %.0f (ideal MFMA-only %.0f per block alone)" % (np.median(cyc), 9216 / 4 * 64 if taps == 9 else 1024 / 4 * 64))
# concurrency over time
st = (d[:, 2] - t0) / 100e6 * 1e6; en = (d[:, 3] - t0) / 100e6 * 1e6
for t in np.linspace(0, en.max(), 12):
    print("  t=%7.1f us  resident blocks %d" % (t, int(((st <= t) & (en > t)).sum())))
