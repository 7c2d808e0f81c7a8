cd $GRAFT_REPO_ROOT; O=$PWD/gpurun_out; mkdir -p $O
{
for m in 1 2 3 1 2; do
  BMC_WGRAD_MERGE=$m BMC_WGRAD_MERGE_MAX_PIXELS=4000000 timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16x6 --also none > $O/r06l_merge$m.json 2> $O/r06l_merge$m.err
  echo "merge $m: $(grep -o '"ms_per_step": [0-9.]*' $O/r06l_merge$m.json | head -1)"
done
} > $O/r06l.log 2>&1
cat $O/r06l.log
