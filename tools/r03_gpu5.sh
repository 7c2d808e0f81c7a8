#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd $R
rm -f $O/r03f_kbench.log
for lib in libbmc_hip.so libbmc_hip_wabl2.so libbmc_hip_wabl6.so libbmc_hip_wabl22.so libbmc_hip_wabl54.so libbmc_hip_wabl118.so libbmc_hip_wabl254.so; do
  echo "== $lib" >> $O/r03f_kbench.log
  BMC_HIP_LIB=$R/bmcnet-esr_amd/csrc/$lib KB_ITERS=100 python tools/kbench.py conv3 2>&1 | grep -v amdgpu >> $O/r03f_kbench.log
done
cat $O/r03f_kbench.log
