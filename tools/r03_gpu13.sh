#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd $R
( timeout 600 python -m pytest tests/test_gpu_r3.py -x -q -m gpu -k "winograd" 2>&1 | tail -3 ) > $O/r03n.log 2>&1
KB_ITERS=100 python tools/kbench.py conv3 2>&1 | grep -v amdgpu >> $O/r03n.log
KB_ITERS=100 KB_B=16 python tools/kbench.py conv3 2>&1 | grep -v amdgpu >> $O/r03n.log
cat $O/r03n.log
