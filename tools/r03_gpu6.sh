#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd $R
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --also none > $O/r03g_bench.json 2> $O/r03g_bench.err
( timeout 1700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_r2.py tests/test_gpu_r3.py tests/test_isa_hygiene.py -q -m gpu 2>&1 | tail -12 ) > $O/r03g_tests.log 2>&1
head -c 400 $O/r03g_bench.json; echo; tail -5 $O/r03g_tests.log
