#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_r3.py -q -x -k "weight_gradient" 2>&1 | tail -2
for lib in bmcnet-esr_amd/csrc/libbmc_hip.so $(ls bmcnet-esr_amd/csrc/libbmc_hip_ww_*.so); do
  echo "== $lib"
  BMC_HIP_LIB=$PWD/$lib timeout 300 python tools/time_wgrad.py 2>&1 | grep -v amdgpu | head -2
done
