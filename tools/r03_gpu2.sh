#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd $R
( timeout 1700 python -m pytest tests/test_gpu_r3.py tests/test_gpu_r2.py tests/test_gpu_parity.py -q -m gpu -s -k "r3 or config3_eventzoom_bf16 or events or voxel or encoder or collate" 2>&1 | grep -v "^$" | cut -c1-2500 | tail -60 ) > $O/r03c_tests.log 2>&1
python bench.py --steps 3 --warmup 1 > $O/r03c_bench.json 2> $O/r03c_bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16x6 --also none > $O/r03c_prof.log 2>&1
find /tmp/ks -name "*kernel_stats.csv" -exec cp {} $O/r03c_kernel_stats.csv \;
cd $R
tail -5 $O/r03c_tests.log; head -c 300 $O/r03c_bench.json; tail -3 $O/r03c_bench.err
