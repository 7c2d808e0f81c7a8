#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p32 -o fp32 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bf16x6 --also none > $O/r03q_prof.log 2>&1
find /tmp/p32 -name "*kernel_stats.csv" -exec cp {} $O/r03q_kernel_stats.csv \;
head -14 $O/r03q_kernel_stats.csv | cut -c1-160
