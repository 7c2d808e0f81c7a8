#!/usr/bin/env python3
"""How long the host needs to ISSUE one C2 training step (no GPU sync in between) vs how long the GPU needs to run it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "bmcnet-esr_amd")):
    sys.path.insert(0, p)
import torch
from models.BMCNet import BMCNet
from train_step import bptt_step, encode_sequence, synthetic_events
dev = torch.device("cuda:0")
B, L, H, W, scale, n_c = 4, 9, 180, 240, 4, 128
torch.manual_seed(0)
m = BMCNet(scale, n_c, 5).to(dev)
opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-5, amsgrad=True)
ev = synthetic_events(B, L, H, W, scale, 24576, dev)
def step():
    inp, gt = encode_sequence(ev, B, L, H, W, scale)
    return bptt_step(m, opt, inp, gt, n_c, scale)
step(); torch.cuda.synchronize()
for _ in range(2):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("host issue %.3f s, until GPU done %.3f s" % (t1 - t0, t2 - t0))
if os.environ.get("HOST_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable(); step(); pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
