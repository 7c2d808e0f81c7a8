import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "bmcnet-esr_amd"))
import torch
from bmc_hip import ops, lib
from bmc_hip.ops import _src, pgemm_raw
dev = torch.device("cuda:0")
B, H, W, C = 16, 180, 240, 128
a = torch.randn(B, H, W, C, device=dev); x = torch.randn(B, H, W, C, device=dev)
fn = lambda: pgemm_raw(_src(a, 0, C, 0, None, 0, B), [_src(x, 0, C, 0, None, 0, B)], B, H, W, 1, 1, C, C, dev)
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): fn()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
print("pgemm<1> 16 x 180x240 x 128x128: %.1f us, %.2f TB/s read, %.1f TFLOP/s" % (ms * 1e3, 2 * a.numel() * 4 / ms / 1e9, 2.0 * B * H * W * C * C / ms / 1e9))
