import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "bmcnet-esr_amd"); sys.path.insert(0, ".")
import torch
from bmc_hip import ops
orig = ops._side_arm
calls = []
def traced(npx):
    st = orig(npx)
    calls.append((npx, None if st is None else (st.side, st.armed)))
    return st
ops._side_arm = traced
import test_gpu_r5 as t
for hw in ((88, 92), (88, 96)):
    calls.clear()
    r = t._two_window_step_vs_oracle(*hw)
    print(hw, "side", r[4], "first calls", calls[:6], "n", len(calls))
