#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd $R
rm -f $O/r03m.log
for mt in 512 200 100 50; do
  echo "== WINO_MIN_TILES=$mt" >> $O/r03m.log
  BMC_WINO_MIN_TILES=$mt python tools/host_time_small.py 2>&1 | grep -v amdgpu | head -3 | tail -1 >> $O/r03m.log
  BMC_WINO_MIN_TILES=$mt HT_H=45 HT_W=80 HT_B=2 python tools/host_time_small.py 2>&1 | grep -v amdgpu | head -3 | tail -1 >> $O/r03m.log
done
cat $O/r03m.log
