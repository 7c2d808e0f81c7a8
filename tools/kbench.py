#!/usr/bin/env python3
"""Isolated-kernel timing at the C2 shapes (batch 2B = 8, 180x240, n_c = 128); HIP events on torch's stream.
usage: python tools/kbench.py [case ...]   cases: conv3 wgrad3 conv1 conv1x256 conv1x256res wgrad1 gram apply relu ln"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bmcnet-esr_amd"))
import torch
from bmc_hip import lib, ops
from bmc_hip.ops import ConvSpec, View

dev = torch.device("cuda:0")
B, H, W, Cn = int(os.environ.get("KB_B", 8)), int(os.environ.get("KB_H", 180)), int(os.environ.get("KB_W", 240)), 128
ITERS = int(os.environ.get("KB_ITERS", 40))


def timeit(fn, flops, name, iters=ITERS):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print("%-28s %9.3f ms  %8.2f TFLOP/s  (%.1f%% of 157.3)" % (name, ms, flops / ms / 1e9, flops / ms / 1e9 / 1.573))


x = torch.randn(B, H, W, Cn, device=dev)
x2 = torch.randn(B, H, W, Cn, device=dev)
g = torch.randn(B, H, W, Cn, device=dev)
npx = B * H * W
cases = sys.argv[1:] or ["conv3", "wgrad3", "conv1", "conv1x256", "wgrad1", "gram", "apply", "relu", "ln"]
s1, s2 = ConvSpec.dense(Cn), ConvSpec.dense(Cn, Cn)
with torch.no_grad():
    if "conv3" in cases:
        w = torch.randn(Cn, Cn, 3, 3, device=dev) * 0.03; b = torch.zeros(Cn, device=dev)
        timeit(lambda: ops.conv([View(x)], w, b, s1, relu=True), 2.0 * npx * Cn * 9 * Cn, "conv3x3 128->128 fwd")
    if "wgrad3" in cases:
        def f():
            sl, ns, G = ops.pgemm_raw(ops._src(g, 0, Cn, 0, None, 0, B), [ops._src(x, 0, Cn, 0, None, 0, B)], B, H, W, 9, B, Cn, Cn, dev)
        timeit(f, 2.0 * npx * Cn * 9 * Cn, "wgrad3x3 128x128 (pgemm)")
        def f2():
            sl, ns, G = ops.pgemm_raw(ops._src(g, 0, Cn, 0, None, 0, B), [ops._src(x, 0, Cn, 0, None, 0, B)], B, H, W, 9, B, Cn, Cn, dev)
            dw = torch.empty(Cn * Cn * 9, device=dev)
            lib.call(lib._red_w, "red", sl.data_ptr(), ns, G, 9, Cn, Cn, s1.kmap(dev).data_ptr(), Cn, dw.data_ptr(), 0, ops._stream())
        timeit(f2, 2.0 * npx * Cn * 9 * Cn, "wgrad3x3 + reduce")
    if "conv1" in cases:
        w = torch.randn(Cn, Cn, 1, 1, device=dev) * 0.1; b = torch.zeros(Cn, device=dev)
        timeit(lambda: ops.conv([View(x)], w, b, s1), 2.0 * npx * Cn * Cn, "conv1x1 128->128 fwd")
    if "conv1x256" in cases:
        w = torch.randn(Cn, 2 * Cn, 1, 1, device=dev) * 0.1; b = torch.zeros(Cn, device=dev)
        timeit(lambda: ops.conv([View(x), View(x2)], w, b, s2), 2.0 * npx * Cn * 2 * Cn, "conv1x1 256->128 fwd")
    if "conv1x256res" in cases:      # the K = 256 1x1 launch with a residual operand (BIE unclustering): conv1p_kernel<16> with an epilogue load
        w = torch.randn(Cn, 2 * Cn, 1, 1, device=dev) * 0.1; b = torch.zeros(Cn, device=dev)
        x3 = torch.randn_like(x)
        timeit(lambda: ops.conv([View(x), View(x2)], w, b, s2, residual=View(x3)), 2.0 * npx * Cn * 2 * Cn, "conv1x1 256->128 + residual")
        timeit(lambda: ops.conv([View(x), View(x2)], w, b, s2, residual=View(x3), relu=True), 2.0 * npx * Cn * 2 * Cn, "conv1x1 256->128 + residual + relu")
    if "wgrad1" in cases:
        def f():
            ops.pgemm_raw(ops._src(g, 0, Cn, 0, None, 0, B), [ops._src(x, 0, Cn, 0, None, 0, B)], B, H, W, 1, B, Cn, Cn, dev)
        timeit(f, 2.0 * npx * Cn * Cn, "wgrad1x1 128x128 (pgemm)")
    if "gram" in cases:
        timeit(lambda: ops.gram(x, x2, 0.088), 2.0 * npx * Cn * Cn, "gram (G=B) + reduce")
    if "apply" in cases:
        p = torch.softmax(torch.randn(B, Cn, Cn, device=dev), -1)
        timeit(lambda: ops.attn_apply(p, x, residual=View(x2)), 2.0 * npx * Cn * Cn, "attn apply (per-sample 1x1)")
    if "relu" in cases:
        timeit(lambda: ops.relu_bwd(g, x), npx * Cn * 12 / 1e3 * 1e3, "relu_bwd (bytes as flops)")
    if "ln" in cases:
        gm = torch.ones(Cn, device=dev); bt = torch.zeros(Cn, device=dev)
        timeit(lambda: ops.layer_norm(x, gm, bt), npx * Cn * 8, "layernorm fwd (bytes as flops)")
