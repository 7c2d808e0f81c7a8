#!/bin/bash
# small-frame step under different switch settings: bash tools/c3_sweep.sh "<env assignments>" ["<env assignments>" ...]   (C3_H, C3_W, C3_B, C3_MATH)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for e in "$@"; do
  for rep in 1 2; do
    r=$(env $e python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-bf16x6 --also none --height ${C3_H:-31} --width ${C3_W:-56} --batch ${C3_B:-4} --math ${C3_MATH:-fp32} 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)
    echo "[$e] ${C3_H:-31}x${C3_W:-56} bs${C3_B:-4} ${C3_MATH:-fp32}: $r"
  done
done
