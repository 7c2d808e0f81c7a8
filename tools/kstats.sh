#!/bin/bash
# usage: tools/kstats.sh <tag> <bench args...>   -> gpurun_out/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats)
tag=$1; shift
R=$PWD; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -o $tag -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bf16x6 "$@" > $R/gpurun_out/${tag}_prof.log 2>&1
find /tmp/ks_$tag -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${tag}_kernel_stats.csv \;
