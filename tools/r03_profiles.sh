#!/bin/bash
# Produces the round's judged artefacts under gpurun_out/r03/ (copied to profiles/ afterwards).
set -u
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
export TMPDIR=/tmp
cd $R
python bench.py > $O/r03_bench_default.json 2> $O/default.err
python bench.py --math bf16x6 --also none --no-cpu-baseline > $O/r03_bench_bf16x6.json 2>> $O/default.err
BMC_WINO=0 python bench.py --also none --no-cpu-baseline --no-bf16x6 > $O/r03_bench_direct_conv.json 2>> $O/default.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p32 -o fp32 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bf16x6 --also none > $O/prof32.log 2>&1
find /tmp/p32 -name "*kernel_stats.csv" -exec cp {} $O/r03_bench_fp32_kernel_stats.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pbf -o bf -- python3 $R/bench.py --steps 3 --warmup 1 --math bf16x6 --no-cpu-baseline --no-bf16x6 --also none > $O/profbf.log 2>&1
find /tmp/pbf -name "*kernel_stats.csv" -exec cp {} $O/r03_bench_bf16x6_kernel_stats.csv \;
cd $R
bash tools/r03_pmc.sh > $O/pmc.log 2>&1
cp $R/gpurun_out/r03_pmc_summary.json $O/ 2>/dev/null
ls -la $O
