#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_r3.py -q -x -k "weight_gradient" 2>&1 | tail -5 > $O/r03n_tests.log
cat $O/r03n_tests.log
timeout 300 python tools/time_wgrad.py 2>&1 | grep -v amdgpu | tail -4
TW_B=8 timeout 300 python tools/time_wgrad.py 2>&1 | grep -v amdgpu | tail -4
TW_B=16 TW_H=31 TW_W=56 timeout 300 python tools/time_wgrad.py 2>&1 | grep -v amdgpu |  tail -4
