#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd $R
( echo "== side0"; BMC_WGRAD_STREAM=0 HOST_PROFILE=1 python tools/host_time_small.py 2>&1 | grep -v amdgpu | head -60
  echo "== side0 bf16"; BMC_WGRAD_STREAM=0 HT_MATH=bf16 python tools/host_time_small.py 2>&1 | grep -v amdgpu | head -5
  echo "== auto"; python tools/host_time_small.py 2>&1 | grep -v amdgpu | head -5 ) > $O/r03k_host.log 2>&1
cat $O/r03k_host.log | cut -c1-150
