#!/usr/bin/env python3
"""cProfile of the host side of one training step at a small frame size (where the step is launch-bound).
python tools/host_profile.py [H W B]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bmcnet-esr_amd")]
import torch

from models.BMCNet import BMCNet
from train_step import bptt_step, encode_sequence, synthetic_events

H, W, B = (int(v) for v in (sys.argv[1:4] + ["31", "56", "4"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = BMCNet(4, 128, 5).to(dev)
opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-5, amsgrad=True)
ev = synthetic_events(B, 9, H, W, 4, 1024, dev)


def step():
    inp, gt = encode_sequence(ev, B, 9, H, W, 4)
    return bptt_step(m, opt, inp, gt, 128, 4)


for _ in range(2):
    step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(3):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host issue %.1f ms/step, wall %.1f ms/step" % ((t1 - t0) / 3 * 1e3, (t2 - t0) / 3 * 1e3))
pr = cProfile.Profile()
pr.enable()
step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)

# ---- the backward pass runs on the autograd engine's device thread: profile it from inside (a hook on the loss enables a second
# profiler on that thread, an end-of-pass callback disables it)
import train_step as TS
pr2 = cProfile.Profile()
orig_backward = torch.Tensor.backward


def backward_profiled(self, *a, **k):
    def on(g):
        pr2.enable()
        torch.autograd.Variable._execution_engine.queue_callback(pr2.disable)
        return g
    self.register_hook(on)
    return orig_backward(self, *a, **k)


torch.Tensor.backward = backward_profiled
step()
torch.Tensor.backward = orig_backward
torch.cuda.synchronize()
print("==== backward pass, engine thread")
pstats.Stats(pr2).sort_stats("tottime").print_stats(40)
