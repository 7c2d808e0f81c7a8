#!/bin/bash
# Two-stream timeline of the C2 step: rocprofv3 kernel trace of a short bench run -> tools/two_stream_timeline.py
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/c2tl -o t -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-bf16x6 --also none "$@" > $O/c2_timeline_bench.log 2>&1
T=$(find /tmp/c2tl -name '*kernel_trace.csv' | head -1)
python3 $R/tools/two_stream_timeline.py $T | tail -12
