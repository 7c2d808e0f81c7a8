#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc3 -o c3 -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-bf16x6 --also none --height 31 --width 56 > $O/r03r_prof.log 2>&1
find /tmp/pc3 -name "*kernel_stats.csv" -exec cp {} $O/r03r_c3_kernel_stats.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc3b -o c3b -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-bf16x6 --also none --height 31 --width 56 --math bf16 > $O/r03r_profb.log 2>&1
find /tmp/pc3b -name "*kernel_stats.csv" -exec cp {} $O/r03r_c3b_kernel_stats.csv \;
grep -o '"ms_per_step": [0-9.]*' $O/r03r_prof.log $O/r03r_profb.log
