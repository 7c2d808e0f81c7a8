#!/bin/bash
# Small-frame profile: rocprofv3 kernel trace + stats of the 31x56 bs 4 step (C3_MATH = fp32 | bf16), timeline + top kernels.
#   usage (GPU box): bash tools/c3_profile.sh <tag> [fp32|bf16] [extra bench flags]
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
TAG=$1; M=${2:-fp32}; shift; shift
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c3p_$M -o t -- python3 $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-bf16x6 --also none --height ${C3_H:-31} --width ${C3_W:-56} --batch ${C3_B:-4} --math $M "$@" > $O/${TAG}_c3_$M.log 2>&1
echo "config3 $M: $(grep -o '"ms_per_step": [0-9.]*' $O/${TAG}_c3_$M.log | head -1)"
python3 $R/tools/gpu_timeline.py $(find /tmp/c3p_$M -name '*kernel_trace.csv' | head -1) 3
S=$(find /tmp/c3p_$M -name '*kernel_stats.csv' | head -1); cp $S $O/${TAG}_c3_${M}_kernel_stats.csv
python3 - $S <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print("all kernels: %.1f ms in %d calls over 6 steps -> %.2f ms, %d launches per step" % (tot / 1e6, calls, tot / 6e6, calls // 6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print("%6d x %7.1f us = %6.2f ms/step  %s" % (int(r["Calls"]) // 6, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 6e6, r["Name"][:110]))
PY
