#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd $R
( timeout 900 python -m pytest tests/test_gpu_r3.py -x -q -m gpu -k "winograd" 2>&1 | tail -3 ) > $O/r03e_wino_tests.log 2>&1
rm -f $O/r03e_kbench.log
for wn in 1 0; do
  echo "== BMC_WINO=$wn" >> $O/r03e_kbench.log
  BMC_WINO=$wn KB_ITERS=100 python tools/kbench.py conv3 >> $O/r03e_kbench.log 2>&1
done
tail -3 $O/r03e_wino_tests.log; cat $O/r03e_kbench.log | grep -v amdgpu.ids
