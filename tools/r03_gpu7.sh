#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd $R
rm -f $O/r03h_kbench.log
for v in 2 1; do
  echo "== BMC_WINO_V=$v" >> $O/r03h_kbench.log
  ( BMC_WINO_V=$v timeout 600 python -m pytest tests/test_gpu_r3.py -x -q -m gpu -k "winograd" 2>&1 | tail -3 ) >> $O/r03h_kbench.log 2>&1
  BMC_WINO_V=$v KB_ITERS=100 python tools/kbench.py conv3 2>&1 | grep -v amdgpu >> $O/r03h_kbench.log
  BMC_WINO_V=$v KB_ITERS=100 KB_B=16 python tools/kbench.py conv3 2>&1 | grep -v amdgpu >> $O/r03h_kbench.log
done
cat $O/r03h_kbench.log
