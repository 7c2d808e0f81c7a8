run() { # label, env side, args...
  l=$1; v=$2; shift 2
  echo "$l side=$v: $(BMC_WGRAD_STREAM=$v timeout 900 python bench.py --no-cpu-baseline --no-bf16x6 --also none "$@" 2>&1 | grep -o "\"ms_per_step\": [0-9.]*\|peak_mem_GiB\": [0-9.]*\|Error.*\|error.*" | head -2 | tr "\n" " ")"
}
for v in 1 0; do run "C2 fp32" $v --steps 8 --warmup 3; done
for v in 0 1; do run "C2 bf16x6" $v --steps 5 --warmup 2 --math bf16x6; done
for v in 0 1; do run "31x56 fp32" $v --steps 20 --warmup 5 --height 31 --width 56; done
for v in 0 1; do run "31x56 bf16" $v --steps 20 --warmup 5 --height 31 --width 56 --math bf16; done
for v in 0 1; do run "45x80 b2 fp32" $v --steps 20 --warmup 5 --height 45 --width 80 --batch 2; done
for v in 0 1; do run "cfg4 recompute" $v --steps 2 --warmup 1 --height 180 --width 190 --batch 8 --seql 17 --recompute; done
for v in 0 1; do run "90x120 fp32" $v --steps 10 --warmup 3 --height 90 --width 120; done
