#!/usr/bin/env python3
"""Isolated timing of bmc_pgemm_reduce_weight (slab reduction of the pixel-reduction GEMM) with and without bias slabs.
usage: python tools/time_reduce.py [nsplit M N taps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bmcnet-esr_amd"))
import torch
from bmc_hip import lib, ops

dev = torch.device("cuda:0")
cases = [tuple(int(v) for v in sys.argv[1:5])] if len(sys.argv) >= 5 else [(224, 128, 128, 1), (256, 128, 128, 1), (64, 128, 128, 1), (256, 128, 256, 1), (28, 128, 128, 9), (64, 128, 64, 9)]
for nsplit, M, N, taps in cases:
    slabs = torch.randn(nsplit * taps * M * N, device=dev)
    bsl = torch.randn(nsplit * 4 * M, device=dev)
    dw = torch.zeros(M * N * taps, device=dev)
    db = torch.zeros(M, device=dev)
    for name, b, d in (("no bias", None, None), ("bias", bsl.data_ptr(), db.data_ptr())):
        fn = lambda: lib.call(lib._red_w, "bmc_pgemm_reduce_weight", slabs.data_ptr(), nsplit, 1, taps, M, N, None, N, dw.data_ptr(), 1, b, d, ops._stream())
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print("nsplit %4d  M %4d N %4d taps %d  %-8s %7.2f us   (%.1f MB of slabs)" % (nsplit, M, N, taps, name, e0.elapsed_time(e1) * 5, slabs.numel() * 4 / 1e6))
