#!/usr/bin/env python3
"""Host issue time vs GPU time of one small-frame training step (31x56 bs4 by default: HT_H / HT_W / HT_B / HT_MATH)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "bmcnet-esr_amd")):
    sys.path.insert(0, p)
import torch
from bmc_hip import ops
from models.BMCNet import BMCNet
from train_step import bptt_step, encode_sequence, synthetic_events
dev = torch.device("cuda:0")
B, L, H, W, scale, n_c = int(os.environ.get("HT_B", 4)), 9, int(os.environ.get("HT_H", 31)), int(os.environ.get("HT_W", 56)), 4, 128
ops.set_math(os.environ.get("HT_MATH", "fp32"))
torch.manual_seed(0)
m = BMCNet(scale, n_c, 5).to(dev)
opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-5, amsgrad=True)
ev = synthetic_events(B, L, H, W, scale, 2048, dev)
def step():
    inp, gt = encode_sequence(ev, B, L, H, W, scale)
    return bptt_step(m, opt, inp, gt, n_c, scale)
for _ in range(4):
    step()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("host issue %.1f ms, until GPU done %.1f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
if os.environ.get("HOST_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable(); step(); pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(30)
