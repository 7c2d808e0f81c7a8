#!/bin/bash
# Produces the round's judged artefacts under gpurun_out/r02/ (copied to profiles/ afterwards).
set -u
R=$PWD; O=$R/gpurun_out/r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
python bench.py > $O/r02_bench_default.json 2> $O/default.err
python bench.py --height 31 --width 56 --math bf16 --warmup 3 --steps 10 > $O/r02_bench_config3_bf16.json 2>> $O/default.err
python bench.py --height 31 --width 56 --math bf16 --graph --warmup 3 --steps 10 > $O/r02_bench_config3_bf16_graph.json 2>> $O/default.err
python bench.py --height 31 --width 56 --warmup 3 --steps 10 > $O/r02_bench_config3_fp32.json 2>> $O/default.err
python bench.py --height 31 --width 56 --graph --warmup 3 --steps 10 > $O/r02_bench_config3_fp32_graph.json 2>> $O/default.err
python bench.py --height 180 --width 190 --seql 17 --batch 8 --recompute --steps 3 --warmup 1 > $O/r02_bench_config4.json 2>> $O/default.err
python bench.py --math bf16x6 > $O/r02_bench_bf16x6.json 2>> $O/default.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p32 -o fp32 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bf16x6 > $O/prof32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pbf -o bf -- python3 $R/bench.py --steps 3 --warmup 1 --math bf16x6 --no-cpu-baseline --no-bf16x6 > $O/profbf.log 2>&1
find /tmp/p32 -name "*kernel_stats.csv" -exec cp {} $O/r02_bench_fp32_kernel_stats.csv \;
find /tmp/pbf -name "*kernel_stats.csv" -exec cp {} $O/r02_bench_bf16x6_kernel_stats.csv \;
ls -la $O
