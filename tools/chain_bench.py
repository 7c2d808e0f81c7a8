#!/usr/bin/env python3
"""Isolated timing of the fused centre chain (csrc/chain.hip) against the three / five launches it replaces, at the C2
shapes (gBIE: 2n = 8 launch batches, lBIE: 2n = 16).  python tools/chain_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bmcnet-esr_amd")]
import torch

from bmc_hip import bie, lib, ops
from bmc_hip.ops import _dense_spec, _src, _stream

dev = torch.device("cuda:0")
H, W, Cn = int(os.environ.get("H", 180)), int(os.environ.get("W", 240)), 128
torch.manual_seed(0)
wf = torch.nn.Parameter(torch.randn(Cn, 2 * Cn, 1, 1, device=dev) * 0.06)
wc = torch.nn.Parameter(torch.randn(Cn, Cn, 1, 1, device=dev) * 0.09)
bf, bc, gamma, beta = (torch.randn(Cn, device=dev) * 0.1 for _ in range(4))
gamma = gamma + 1


def timeit(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


for n in (4, 8):
    B2 = 2 * n
    xs = torch.randn(n, H, W, Cn, device=dev)
    x12 = torch.randn(B2, H, W, Cn, device=dev)
    dc = torch.randn(B2, H, W, Cn, device=dev)
    gx = torch.randn(n, H, W, Cn, device=dev)
    X = lambda t, **k: _src(t, 0, Cn, k.get("shift", 0), k.get("mod"), k.get("b0", 0), k.get("B", t.shape[0]))
    flop = 2.0 * B2 * H * W * Cn * 3 * Cn
    fwd = lambda: bie.chain_fwd(X(xs, mod=n, B=B2), X(x12, shift=n, mod=B2), wf, bf, gamma, beta, wc, bc, 1e-6, B2, H, W, Cn, dev)
    yhat, rstd, c = fwd()
    bwd = lambda: bie.chain_bwd(X(dc), yhat, rstd, gamma, wf, wc, X(gx), n, H, W, Cn, dev)
    tf, tb = timeit(fwd), timeit(bwd)
    print("2n=%2d  chain fwd %.3f ms (%.1f TF, %.0f GB/s)   chain bwd %.3f ms (%.1f TF)" %
          (B2, tf, flop / tf / 1e9, B2 * H * W * 2052 / tf / 1e6, tb, flop / tb / 1e9), flush=True)
    # the launches it replaces (forward): conv 2C->C, LayerNorm, conv C->C
    s1, s2 = _dense_spec(Cn), bie._spec2(Cn)
    z, y, cc = (torch.empty(B2, H, W, Cn, device=dev) for _ in range(3))
    st = torch.empty(B2 * H * W * 2, device=dev)

    def unfused():
        bie._conv([X(xs, mod=n, B=B2), X(x12, shift=n, mod=B2)], wf.detach().reshape(1, Cn, 2 * Cn, 1), s2, wf, bf, z, B2)
        lib.call(lib._ln_fwd, "ln", z.data_ptr(), gamma.data_ptr(), beta.data_ptr(), B2 * H * W, Cn, 1e-6, y.data_ptr(), st.data_ptr(), _stream())
        bie._conv([X(y)], wc.detach().reshape(1, Cn, Cn, 1), s1, wc, bc, cc, B2)
    tu = timeit(unfused)
    print("       unfused fwd (conv, LN, conv) %.3f ms (%.1f TF)" % (tu, flop / tu / 1e9), flush=True)
