#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd $R
( timeout 900 python -m pytest tests/test_gpu_r3.py -x -q -m gpu -s -k "winograd" 2>&1 | grep -v "^$" | cut -c1-1500 | tail -40 ) > $O/r03d_wino_tests.log 2>&1
for wn in 1 0; do
  echo "== BMC_WINO=$wn" >> $O/r03d_kbench.log
  BMC_WINO=$wn KB_ITERS=100 python tools/kbench.py conv3 >> $O/r03d_kbench.log 2>&1
  BMC_WINO=$wn KB_ITERS=100 KB_B=16 python tools/kbench.py conv3 >> $O/r03d_kbench.log 2>&1
done
tail -12 $O/r03d_wino_tests.log; cat $O/r03d_kbench.log | grep -v amdgpu.ids
