import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bmcnet-esr_amd"))
import torch, numpy as np
from bmc_hip import lib, ops
dev = torch.device("cuda:0")
B, H, W, Cn = 8, 180, 240, 128
taps = int(sys.argv[1]) if len(sys.argv) > 1 else 9
x = torch.randn(B, H, W, Cn, device=dev); g = torch.randn(B, H, W, Cn, device=dev)
gs = ops._src(g, 0, Cn, 0, None, 0, B); xs = ops._src(x, 0, Cn, 0, None, 0, B)
for _ in range(20):
    ops.pgemm_raw(gs, [xs], B, H, W, taps, B, Cn, Cn, dev, want_bias=True)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 4096)()
lib._lib.bmc_pstamp_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib._lib.bmc_pstamp_read(buf, 4096) == 0
a = np.array(buf[:256 * 4], dtype=np.float64).reshape(256, 4)
n = a[:, 3].mean()
print("fp32 pgemm taps=%d per tile (wave 0): issue %.0f  bias+mma %.0f  barrier %.0f cycles; tiles/WG %.1f" % (taps, a[:, 0].mean() / n, a[:, 1].mean() / n, a[:, 2].mean() / n, n))
