#!/usr/bin/env python3
"""Which ATen arithmetic is still launched inside one training step (adds / cats / fills of activation-sized tensors), and
which backward node triggers it: torch.profiler over one small BPTT step, grouped by (op, shape, enclosing autograd node)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bmcnet-esr_amd"))
import torch
from models.BMCNet import BMCNet
from train_step import bptt_step, encode_sequence, synthetic_events
dev=torch.device("cuda:0")
torch.manual_seed(0)
B,H,W,L=2,32,48,4
m=BMCNet(4,32,2).to(dev)
opt=torch.optim.Adam(m.parameters(), lr=1e-4)
ev=synthetic_events(B,L,H,W,4,1024,dev,seed=1)
def step():
    inp,gt=encode_sequence(ev,B,L,H,W,4)
    return bptt_step(m,opt,inp,gt,32,4)
step(); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=False) as prof:
    step(); torch.cuda.synchronize()
cnt=collections.Counter()
def anc(e):
    names=[]
    q=e.cpu_parent
    while q is not None and len(names)<3:
        names.append(q.name[:60]); q=q.cpu_parent
    return " <- ".join(names)
for e in prof.events():
    if e.name in ("aten::add_","aten::add","aten::cat","aten::zero_") and e.input_shapes and e.input_shapes[0] and len(e.input_shapes[0])==4 and e.input_shapes[0][1]==32:
        cnt[(e.name, str(e.input_shapes[0]), anc(e))]+=1
for k,v in sorted(cnt.items(), key=lambda kv:-kv[1])[:30]:
    print(v, k)
