#!/usr/bin/env python3
"""Per-wave cycle stamps of the F(4x4) weight-gradient kernel (diagnostic build: make -C bmcnet-esr_amd/csrc libbmc_hip_w4gstampN.so;
BMC_HIP_LIB=...).  Prints, per wave, the median cycles of every phase of a stage (workgroup 8, iterations 40..103)."""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "bmcnet-esr_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from bmc_hip import lib, ops

dev = torch.device("cuda:0")
B, H, W = int(os.environ.get("TW_B", 8)), 180, 240
x = torch.randn(B, H, W, 128, device=dev)
g = torch.randn(B, H, W, 128, device=dev)
spec = ops.ConvSpec.dense(128)
w = torch.zeros(128, 128, 3, 3, device=dev)
b = torch.zeros(128, device=dev)
ops.WINO4_WGRAD = True
for _ in range(5):
    ops._wgrad_plain(g, x, spec, w, b, 9)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (8 * 64 * 8))()
f = lib._lib.bmc_w4g_read_stamps
f.argtypes = [C.c_void_p]
assert f(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(8, 64, 8).astype(np.int64)
names = ["rq_begin", "early: T+pieces", "multiply", "late: pieces+T", "vmcnt wait", "patch+barrier", "(next top)"]
print("median cycles per phase (wave: " + ", ".join(names[:6]) + ", whole stage)")
for wv in range(8):
    d = np.diff(t[wv, :, :7], axis=1)
    stage = np.diff(t[wv, :, 0])
    print("wave %d: %s   stage %d" % (wv, " ".join("%6d" % int(np.median(d[:, k])) for k in range(6)), int(np.median(stage))))
# relative start of multiply between partners
for wv in range(4):
    print("wave %d vs %d: M starts %+d / ends %+d cycles apart" % (wv, wv + 4, int(np.median(t[wv, :, 2] - t[wv + 4, :, 2])), int(np.median(t[wv, :, 3] - t[wv + 4, :, 3]))))
