import os, sys
sys.path.insert(0, '/root/repo/bmcnet-esr_amd')
import torch
from bmc_hip.ops import _packed_weight, _src, conv_raw, coutpad, ConvSpec
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(404)
Cn = 128; spec = ConvSpec.dense(Cn); cp = coutpad(Cn)
w = (torch.randn(1, Cn, Cn, 9, generator=g) * 0.03).to(dev)
bias = torch.randn(1, Cn, generator=g).to(dev)
B, H, W = (int(v) for v in sys.argv[1:4])
x = torch.randn(B, H, W, Cn, generator=g).to(dev)
outs = []
for wino in (4, 0):
    wp = _packed_weight(w, spec, None, wino=wino)
    out = torch.full((B, H, W, Cn), 7.0, device=dev)
    for rep in range(3):
        conv_raw([_src(x, 0, Cn, 0, None, 0, B)], wp, spec.kpad * 9 * cp, bias, Cn, out.data_ptr(), H * W * Cn, Cn, B, H, W, Cn, 9, relu=False, bpg=B, wino=wino)
    outs.append(out)
d = (outs[0] - outs[1]).abs().amax(-1)          # [B,H,W]
print("split", os.environ.get("BMC_W4_SPLIT", "1"), "rel", float((outs[0]-outs[1]).norm()/outs[1].norm()))
tx = (W + 3) // 4; ty = (H + 3) // 4
bad = (d > 1e-3)
for b in range(B):
    if bad[b].any():
        ys, xs = torch.nonzero(bad[b], as_tuple=True)
        tiles = sorted(set(((ys // 4) * tx + xs // 4).tolist()))
        wt = sorted(set(t // 16 for t in tiles))
        print(" image", b, "bad pixels", int(bad[b].sum()), "wg tiles", wt, "global wg tile", [b * ((tx*ty+15)//16) + t for t in wt])
