#!/bin/bash
# Ablation ladder of the F(4x4) weight-gradient kernel: make -C bmcnet-esr_amd/csrc libbmc_hip_w4gablN.so first (bits: BMC_W4G_ABL).
R=${GRAFT_REPO_ROOT:-$PWD}
for n in ${W4G_ABLS:-0 1 2 4 8 6 14 15}; do
  lib=$R/bmcnet-esr_amd/csrc/libbmc_hip_w4gabl$n.so; [ $n = 0 ] && lib=$R/bmcnet-esr_amd/csrc/libbmc_hip.so
  [ -f $lib ] && echo "abl $n: $(TW_B=${TW_B:-8} TW_ONLY=winograd4 BMC_HIP_LIB=$lib timeout 120 python $R/tools/time_wgrad.py 2>&1 | grep 'main kernel')"
done
