#!/usr/bin/env python3
"""Per-window latency of streaming inference (the reference's `time` metric, infer_BMCNet.py:44-68): eager vs HIP-graph
replay, BMCNet(4,128,5), batch 1.  python tools/infer_latency.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bmcnet-esr_amd")]
import torch

from infer import StreamingSR
from models.BMCNet import BMCNet

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = BMCNet(4, 128, 5).to(dev)
for H, W in ((45, 80), (31, 56), (180, 240)):
    frames = torch.poisson(torch.full((1, 40, 2, H, W), 0.284)).to(dev)
    for graph in (False, True):
        sr = StreamingSR(m, 128, 4, graph=graph)
        for i in range(30):
            sr.step(frames[:, i:i + 2].transpose(1, 2))
        print("%3dx%-3d  %-6s  %.2f ms/window (%.1f windows/s)" % (H, W, "graph" if graph else "eager", sr.latency_ms(skip=5), 1e3 / sr.latency_ms(skip=5)))
