"""CPU checks of the Winograd arithmetic behind csrc/wino.hip (F(2x2,3x3)) and csrc/wino4.hip (F(4x4,3x3)), on the float32 emulation
of tests/wino_numerics.py (transforms, contraction and inverse transform as the kernels compute them):
  * the transform matrices are exact minimal-filtering algorithms (float64 identity);
  * one 3x3 convolution through F(4x4) stays within the error the round-4 decision was based on (profiles/r04_wino_numerics.txt);
  * the exact-zero property that the routing rule of bmc_hip.ops.exact_zero_inputs rests on: F(2x2) gives EXACTLY 0 where a pixel's
    3x3 field holds no input (every output is a combination of products of its own field), F(4x4) does not."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import wino_numerics as WN  # noqa: E402


def test_transform_matrices_are_exact_minimal_filtering_algorithms():
    rng = np.random.default_rng(1)
    for m, mats in ((2, WN.lavin(2)), (4, WN.lavin(4)), (4, WN.cook_toom([0, 1, -1, .5, -2], 4)), (3, WN.cook_toom([0, 1, -1, 2], 3))):
        AT, G, BT = mats
        n = m + 2
        for _ in range(5):
            d, g = rng.standard_normal((n, n)), rng.standard_normal((3, 3))
            y = AT @ ((G @ g @ G.T) * (BT @ d @ BT.T)) @ AT.T
            ref = np.array([[(d[i:i + 3, j:j + 3] * g).sum() for j in range(m)] for i in range(m)])
            assert np.abs(y - ref).max() < 1e-10


def test_f4x4_convolution_error_is_what_the_decision_assumed():
    torch.manual_seed(0)
    B, C, K, H, W = 1, 64, 64, 23, 38                   # ragged: 6 x 10 tiles, last row / column partial
    x = torch.relu(torch.randn(B, C, H, W))
    w = torch.randn(K, C, 3, 3) * (2.0 / (9 * C)) ** 0.5
    g = torch.randn(B, K, H, W)
    y64 = F.conv2d(x.double(), w.double(), None, padding=1)
    dw64 = torch.nn.grad.conv2d_weight(x.double(), w.shape, g.double(), padding=1)
    s2, s4 = WN.Scheme("F2", 2, WN.lavin(2)), WN.Scheme("F4", 4, WN.lavin(4))
    e2, e4 = WN.rel(WN.wino_fwd(x, w, s2), y64), WN.rel(WN.wino_fwd(x, w, s4), y64)
    assert e2 < 1e-6 and e4 < 5e-6 and e4 < 12 * e2, (e2, e4)
    assert WN.rel(WN.wino_wgrad(x, g, s2), dw64) < 2e-6


def test_exact_zero_fields_f2x2_keeps_them_f4x4_does_not():
    g = torch.Generator().manual_seed(3)
    B, C, K, H, W = 1, 16, 16, 32, 48
    x = torch.poisson(torch.full((B, C, H, W), 0.05), generator=g) * (torch.rand(B, 1, H, W, generator=g) < 0.1)
    w = torch.randn(K, C, 3, 3, generator=g) * 0.1
    occupied = F.max_pool2d(x.abs().sum(1, keepdim=True), 3, 1, 1) > 0
    empty = (~occupied).expand(B, K, H, W)
    assert empty.float().mean() > 0.2
    y2 = WN.wino_fwd(x, w, WN.Scheme("F2", 2, WN.lavin(2)))
    y4 = WN.wino_fwd(x, w, WN.Scheme("F4", 4, WN.lavin(4)))
    yd = F.conv2d(x, w, None, padding=1)
    assert torch.count_nonzero(yd[empty]) == 0
    assert torch.count_nonzero(y2[empty]) == 0                       # exact zeros survive F(2x2)
    assert torch.count_nonzero(y4[empty]) > 0                        # ... and do not survive F(4x4): hence ops.exact_zero_inputs
    assert WN.rel(y4, yd.double()) < 1e-5
