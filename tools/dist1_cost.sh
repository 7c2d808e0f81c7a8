cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for mode in plain dist; do
  if [ $mode = dist ]; then export BMC_FORCE_DIST=1; else unset BMC_FORCE_DIST; fi
  python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-bf16x6 --also none 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode', d['ms_per_step'], d['config']['rccl_ranks'])"
done
cd /tmp
BMC_FORCE_DIST=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dist -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-bf16x6 --also none > /dev/null 2>&1
unset BMC_FORCE_DIST
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_plain -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-bf16x6 --also none > /dev/null 2>&1
for m in dist plain; do f=$(find /tmp/prof_$m -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/r05_${m}1_kernel_stats.csv; echo "== $m"; head -25 $f | cut -d, -f1-4 | cut -c1-150; done
