#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd $R
( timeout 900 python -m pytest tests/test_gpu_r3.py -x -q -m gpu -k "side_stream" 2>&1 | tail -3 ) > $O/r03l.log 2>&1
( echo "== fp32 side0"; BMC_WGRAD_STREAM=0 python tools/host_time_small.py 2>&1 | grep -v amdgpu | head -3
  echo "== fp32 auto"; python tools/host_time_small.py 2>&1 | grep -v amdgpu | head -3
  echo "== bf16 side0"; BMC_WGRAD_STREAM=0 HT_MATH=bf16 python tools/host_time_small.py 2>&1 | grep -v amdgpu | head -3
  echo "== bf16 side1"; BMC_WGRAD_STREAM=1 HT_MATH=bf16 python tools/host_time_small.py 2>&1 | grep -v amdgpu | head -3
  echo "== 45x80 bs2 side0"; BMC_WGRAD_STREAM=0 HT_H=45 HT_W=80 HT_B=2 python tools/host_time_small.py 2>&1 | grep -v amdgpu | head -3
  echo "== 45x80 bs2 auto"; HT_H=45 HT_W=80 HT_B=2 python tools/host_time_small.py 2>&1 | grep -v amdgpu | head -3 ) >> $O/r03l.log 2>&1
for side in 0 1; do for m in fp32 bf16; do
  BMC_WGRAD_STREAM=$side python bench.py --height 31 --width 56 --math $m --graph --steps 10 --warmup 4 --no-cpu-baseline --no-bf16x6 --also none > $O/r03l_c3_${m}_graph_side$side.json 2>> $O/r03l.err
  python -c "
import json; j=json.load(open('$O/r03l_c3_${m}_graph_side$side.json')); print('graph $m side$side', j['ms_per_step'], j['value'])" >> $O/r03l.log
done; done
cat $O/r03l.log
