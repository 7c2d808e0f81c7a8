import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bmcnet-esr_amd"))
import torch, numpy as np
from bmc_hip import lib, ops
from bmc_hip.ops import ConvSpec, View
dev = torch.device("cuda:0")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B, H, W, Cn = 8, 180, 240, 128
x = torch.randn(B, H, W, Cn, device=dev)
w = torch.randn(Cn, Cn, k, k, device=dev) * 0.05; b = torch.zeros(Cn, device=dev)
spec = ConvSpec.dense(Cn)
z = torch.zeros(8192, dtype=torch.int64)
with torch.no_grad():
    for _ in range(3):
        ops.conv([View(x)], w, b, spec)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8192)()
lib._lib.bmc_fstamp_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib._lib.bmc_fstamp_read(buf, 8192) == 0
a = np.array(buf[:768 * 8], dtype=np.float64).reshape(768, 8)
n = a[:, 4].sum()
print("fp32 conv k=%d epilogue per tile (wave 0): setup %.0f  pass1 (loads+math) %.0f  stores %.0f  zero %.0f cycles  (%d epilogues)" % (k, a[:, 0].sum() / n, a[:, 1].sum() / n, a[:, 2].sum() / n, a[:, 3].sum() / n, n))
