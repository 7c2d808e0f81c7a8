#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd $R
( timeout 900 python -m pytest tests/test_gpu_r3.py -x -q -m gpu -s -k "side_stream" 2>&1 | tail -5 ) > $O/r03j_tests.log 2>&1
for side in 0 auto; do
  BMC_WGRAD_STREAM=$side python bench.py --height 31 --width 56 --steps 10 --warmup 4 --no-cpu-baseline --no-bf16x6 --also none > $O/r03j_c3_fp32_side$side.json 2>> $O/r03j.err
  BMC_WGRAD_STREAM=$side python bench.py --height 31 --width 56 --math bf16 --steps 10 --warmup 4 --no-cpu-baseline --also none > $O/r03j_c3_bf16_side$side.json 2>> $O/r03j.err
  BMC_WGRAD_STREAM=$side python bench.py --height 45 --width 80 --batch 2 --steps 10 --warmup 4 --no-cpu-baseline --no-bf16x6 --also none > $O/r03j_nfs_side$side.json 2>> $O/r03j.err
done
tail -3 $O/r03j_tests.log
for f in $O/r03j_*.json; do python -c "
import json,sys; j=json.load(open('$f')); print('$f'.split('/')[-1], j['ms_per_step'], j['value'])"; done
