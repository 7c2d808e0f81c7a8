#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
python bench.py --also none --no-cpu-baseline --no-bf16x6 > $O/r03p_bench.json 2> $O/r03p_bench.err
cat $O/r03p_bench.json | cut -c1-400
timeout 2400 python -m pytest tests -q -x -m gpu 2>&1 | tail -8 > $O/r03p_tests.log
cat $O/r03p_tests.log
