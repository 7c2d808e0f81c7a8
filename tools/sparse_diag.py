import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "bmcnet-esr_amd"); sys.path.insert(0, ".")
import test_gpu_r5 as t
from bmc_hip import ops
for name, w, w4 in (("default", True, True), ("F(2x2) everywhere", True, False), ("direct kernel", False, False)):
    ops.WINO, ops.WINO4 = w, w4
    print("=====", name)
    try:
        t.test_sparse_recording_bias_gradients_vs_oracle("zero")
    except AssertionError as e:
        print("ASSERT", str(e)[:200])
