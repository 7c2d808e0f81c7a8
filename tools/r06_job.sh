cd $GRAFT_REPO_ROOT; O=$PWD/gpurun_out; mkdir -p $O
export BMC_PMC_COMMIT=fa91bc8
{
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
bash tools/gpu_run.sh r06 stats stats1 pmc
} > $O/r06s.log 2>&1
tail -30 $O/r06s.log | cut -c1-300
